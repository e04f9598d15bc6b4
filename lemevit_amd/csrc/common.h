// common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/lemevit_hip.h"

typedef unsigned short bf16_t;  // storage type: raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LMV_WAVE 64

// ---- error plumbing (host) -----------------------------------------------------------------
void lmv_set_error(const char* fmt, ...);
#define LMV_FAIL(code, ...)          \
  do {                               \
    lmv_set_error(__VA_ARGS__);      \
    return (code);                   \
  } while (0)
// ---- the persistent stage kernels' error word (misc.hip): one sticky word per device (pinned host memory, device-mapped) that a bounded spin sets when it runs out (a lost
// hand-off); read by lmv_stage_error_count without a synchronisation.  `stream`: the stream of the launch (a first call inside a stream capture is refused: NULL).
unsigned* lmv_stage_errword(void* stream);
// workgroups of `kernel` the device holds at once (occupancy query x CUs; misc.hip); 0: unknown
int lmv_stage_capacity(const void* kernel, int threads, size_t lds_bytes);

// ---- launch timing probe (lmv_debug_launch_timing, misc.hip): HIP events around the forward-form Linear entry points ON THE STREAM THEY LAUNCH ON, whichever schedule calls
// them (the native block schedule of csrc/block.hip included) -- bench.py's roofline object.  Off: one predictable branch per entry.
extern bool g_lmv_timing_on;
void lmv_timing_begin(void* stream, double flops, double bytes, int kind);
void lmv_timing_end(void* stream);
void lmv_timing_set_kind(int kind);
// launch kinds of the probe = kernels (bench.py names the kind with the largest total as roofline.kernel)
enum { LMV_TK_GEMM_NT = 0, LMV_TK_SSTAGE = 1, LMV_TK_DSTAGE = 2, LMV_TK_STEM = 3, LMV_TK_GEMM_DX = 4, LMV_TK_GEMM_DW = 5, LMV_TK_SPLITK_REDUCE = 6, LMV_TK_ATTN_FWD = 7,
       LMV_TK_RS_GEMM = 8, LMV_TK_WN_GEMM = 9, LMV_TK_ATTN_BWD = 10, LMV_TK_RSW_GEMM = 11 };
struct LmvTimedLaunch {          // RAII bracket of one entry point (or of one launch inside it)
  void* st; bool on;
  LmvTimedLaunch(void* stream, double flops, double bytes, int kind = 0) : st(stream), on(g_lmv_timing_on) { if (on) lmv_timing_begin(st, flops, bytes, kind); }
  void kind(int k) { if (on) lmv_timing_set_kind(k); }
  ~LmvTimedLaunch() { if (on) lmv_timing_end(st); }
};
#define LMV_CHECK_LAUNCH(name)                                                \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// out[map(i)] += sum_r partial[r][i], i < width  (misc.hip).  mode 0: i < na -> out_a[i], else out_b[i - na];
// mode 1 (dwconv, partial index i = tap * C + c with na = C): tap < 9 -> out_a[c * 9 + tap], tap == 9 -> out_b[c].
int lmv_launch_partial_reduce(const float* partial, int nrows, int width, float* out_a, int na, float* out_b, int mode, hipStream_t st);

// Workgroup body of the partial-row reduction (partial_reduce_kernel, misc.hip; also run by the blocks of lmv_reduce_batch that own a
// LMV_REDUCE_ROWS segment, gemm.hip -- ONE summation tree, so the merged launch is bit-identical to the stand-alone one):
// 8 float4 columns (128 B of each partial row) x 32 row-groups per workgroup `blk`; every thread sums rows rg, rg + 32, ... with four
// independent 16-byte loads in flight, then the 32 row-groups are combined through LDS (red: 256 float4).
static __device__ __forceinline__ void lmv_partial_reduce_block(float4* red, int blk, const float* __restrict__ partial, int nrows, int width,
                                                                float* __restrict__ out_a, int na, float* __restrict__ out_b, int mode) {
  const int cx = threadIdx.x & 7, rg = threadIdx.x >> 3, i = (blk * 8 + cx) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < width) {
    const float* p = partial + i;
    int r = rg;
    for (; r + 96 < nrows; r += 128) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (int64_t)r * width);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (int64_t)(r + 32) * width);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (int64_t)(r + 64) * width);
      const float4 v3 = *reinterpret_cast<const float4*>(p + (int64_t)(r + 96) * width);
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; r < nrows; r += 32) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)r * width);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[rg * 8 + cx] = a;
  __syncthreads();
  if (rg == 0 && i < width) {
#pragma unroll 8
    for (int k = 1; k < 32; ++k) { const float4 v = red[k * 8 + cx]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    const float s[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = i + e;
      if (mode == 0) { if (j < na) out_a[j] += s[e]; else out_b[j - na] += s[e]; }
      else { const int t = j / na, ch = j - t * na; if (t < 9) out_a[ch * 9 + t] += s[e]; else out_b[ch] += s[e]; }
    }
  }
}

// ---- A/B switches ------------------------------------------------------------------------------
// Every tuning switch of the library lives here; lmv_config_set(key, value) changes one at run time (tests and the tools/ sweeps use it).  Defaults = measured best.
// Round 6: only the five switches that still name an LMV_* variable below are also read from the environment (ONCE, when the library is loaded -- never on a launch path).
struct LmvConfig {
  int gemm_bk;            //            0 = auto, 32 / 64 force the k-tile depth of the bf16 GEMMs
  int gemm_bk32_tiles;    //    fwd / dX launches with at least this many 128 x 128 tiles use 32-deep k-tiles (4 workgroups per CU)
  int dw_bk;              //              k-tile depth of the weight-gradient GEMM (32)
  int dw_target_blocks;   // LMV_DW_TARGET_BLOCKS   0 = auto: workgroups one generation of a weight-gradient launch should have
  int gemm_no_dma;        //        1: register-staged operand path instead of LDS-DMA
  int gemm_w8;            // LMV_GEMM_W8            1: 8-wave 64-deep forward kernel; 2: also for every dX launch; 0: off
  int gemm_cumap;         //         1: CU-aware tile order
  int gemm_nst, gemm_nst_dw;   // ring depth of the 32-deep loop (2 / 3)
  int gemm_wn;            // LMV_GEMM_WN            1: whole-width kernel (wngemm.hip) for the 384-wide forward-form launches where it measured faster; 0: off; 2: wherever it applies
  int gemm_rs;            // LMV_GEMM_RS            1: register-stationary kernels (rsgemm.hip) where they measured faster; 0: off; 2: wherever they apply
  int dwconv_v;           //           0 = auto: rows per thread of the depth-wise convolution kernels
  int ln_exact_fused;     //     1: norm1 of the C = 96 blocks runs inside its projection launches (lmv_ln_linear_exact_fwd, csrc/rswgemm.hip)
  int res_ln_fused;       //       1: "S" blocks run the attention projection and norm2 as one launch where lmv_linear_res_ln_fwd applies
  int dx_ln_fused;        //        1: lmv_block_bwd fuses the dX of fc1 / qkv with the LayerNorm backward of their input where lmv_linear_dx_ln_bwd applies
  int mlp_split384;       //       1: fused inference schedule runs the C = 384 MLP half as LayerNorm + rsgemm fc1 + wngemm fc2 instead of the one-kernel form
  int mlp_rw96;           //           1: the C = 96 one-kernel MLP with both weight matrices resident in LDS (csrc/rwmlp.hip) instead of the tile-streaming form
  int mlp_tm;             //             0 = auto: token rows per workgroup of the fused MLP kernel (64 / 128)
  int attn_pv16, attn_fuse_dq, attn_fused_bwd, attn_pair;      //  (attn_fused_bwd: 0 separate dQ / dK-dV kernels, 1 round 2's fused backward, 2 (default) round 5's: dQ by query tile, one barrier per block)
  int ln_bwd_blocks, ln_bwd_minrows;                           //
  int stage_ticket_skew;  // LMV_STAGE_TICKET_SKEW  test switch (0): the persistent stage kernels ask ticket counter (XCC_ID + skew * hash(blockIdx)) & 7 first -- a simulated foreign workgroup -> XCD placement
};
LmvConfig& lmv_config();

// rsgemm.hip: the register-stationary GEMM for wide short-reduction forward-form launches (bf16)
bool lmv_rs_eligible(const lmv_linear_problem* p, int nproblems, int N, int K, int act, bool force);
int lmv_rs_linear(const lmv_linear_problem* p, int nproblems, int N, int K, int act, hipStream_t st);
// wngemm.hip: one workgroup per 128-row token panel and ALL 384 output columns (bf16, forward form)
bool lmv_wn_eligible(const lmv_linear_problem* p, int nproblems, int N, int K, int act, bool force);
// csrc/rwmlp.hip: lmv_mlp_fused_fwd at C = 96 / hidden = 384 with both weight matrices resident in LDS
bool lmv_mlp_rw96_eligible(const lmv_mlp_problem* p, int nproblems, int C, int hidden);
int lmv_mlp_rw96_fwd(const lmv_mlp_problem* p, int nproblems, const lmv_mlp_weights* w, float eps, hipStream_t st);
int lmv_wn_linear(const lmv_linear_problem* p, int nproblems, int N, int K, int act, hipStream_t st);

static inline bool lmv_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- bf16 <-> f32 ---------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// f32 -> bf16, round to nearest even, NaN stays NaN: ONE v_cvt_pk_bf16_f32 per pair on gfx950 (the integer recipe with
// its NaN branch costs ~8 VALU ops and an exec-mask branch per element)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte chunk <-> floats
template <typename T> __device__ __forceinline__ void chunk_to_f(const uint4& c, float* f);
template <> __device__ __forceinline__ void chunk_to_f<float>(const uint4& c, float* f) {
  f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y); f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
}
template <> __device__ __forceinline__ void chunk_to_f<bf16_t>(const uint4& c, float* f) {
  f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
  f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
  f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
  f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 f_to_chunk(const float* f);
template <> __device__ __forceinline__ uint4 f_to_chunk<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 f_to_chunk<bf16_t>(const float* f) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

// load / store n (= 4) consecutive elements as floats (8 B for bf16, 16 B for f32)
__device__ __forceinline__ void ld4(const float* p, float* f) {
  float4 v = *reinterpret_cast<const float4*>(p); f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float* f) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float* f) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
__device__ __forceinline__ void st4(bf16_t* p, const float* f) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3])); }

// exact-erf GELU and its derivative (nn.GELU default, models/lemevit.py:528)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * expf(-0.5f * x * x);
}
// bf16 epilogues: branch-free erf (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7 -- far below bf16 resolution) sharing its
// exponential with the Gaussian term of GELU'.  ~12 VALU ops instead of libm erff's ~40 with branches.
__device__ __forceinline__ void erf_as(float x, float* erfv, float* gauss) {   // erf(x / sqrt 2), exp(-x^2 / 2)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float ex = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  *erfv = copysignf(1.0f - poly * ex, x);
  *gauss = ex;
}
__device__ __forceinline__ float gelu_fast_f(float x) { float e, g; erf_as(x, &e, &g); return 0.5f * x * (1.0f + e); }
__device__ __forceinline__ float gelu_grad_fast_f(float x) {
  float e, g; erf_as(x, &e, &g);
  return 0.5f * (1.0f + e) + x * 0.39894228040143268f * g;
}

// GELU of TWO values for the fused MLP kernel (fused.hip), where the activation is the binding VALU cost (4 C values per token, all
// produced on-chip between two MFMA phases): Phi(x) = 0.5 + s q(s^2), s = clamp(x / 4, -1, 1), q = degree-7 minimax fit of
// erf(x / sqrt 2) / 2 on [-4, 4] with q(1) = 1/2 exactly (so GELU(x <= -4) = 0 and GELU(x >= 4) = x).  No transcendental, and every
// step is a packed-fp32 instruction (v_pk_mul_f32 / v_pk_fma_f32: two values per lane and issue).  |Phi error| <= 4.8e-5,
// |GELU error| <= 1.9e-4 absolute (3.2e-5 for |x| <= 2) -- below half a bf16 ulp of the stored activation for |h| >= 0.05, and the
// hidden activations only ever exist as bf16 MFMA operands.  (The stand-alone GEMM epilogues keep the erf formula above.)
// (hipcc SLP-packs a scalar-float version of the same code into the same v_pk_* instructions; with -fno-slp-vectorize the pass is
//  10 - 15 % slower, tools/mlp_timeline.py.)
__device__ __forceinline__ f32x2_t gelu_poly2(f32x2_t x) {
  f32x2_t s = x * 0.25f;
  s[0] = __builtin_amdgcn_fmed3f(s[0], -1.0f, 1.0f); s[1] = __builtin_amdgcn_fmed3f(s[1], -1.0f, 1.0f);
  const f32x2_t u = s * s;
  f32x2_t q = {-1.6300047636032104f, -1.6300047636032104f};
  q = __builtin_elementwise_fma(q, u, f32x2_t{7.93373966217041f, 7.93373966217041f});
  q = __builtin_elementwise_fma(q, u, f32x2_t{-16.877059936523438f, -16.877059936523438f});
  q = __builtin_elementwise_fma(q, u, f32x2_t{20.921268463134766f, 20.921268463134766f});
  q = __builtin_elementwise_fma(q, u, f32x2_t{-17.09065055847168f, -17.09065055847168f});
  q = __builtin_elementwise_fma(q, u, f32x2_t{9.8812894821167f, 9.8812894821167f});
  q = __builtin_elementwise_fma(q, u, f32x2_t{-4.233964920043945f, -4.233964920043945f});
  q = __builtin_elementwise_fma(q, u, f32x2_t{1.595382571220398f, 1.595382571220398f});
  const f32x2_t phi = __builtin_elementwise_fma(s, q, f32x2_t{0.5f, 0.5f});
  return x * phi;
}

// sum / max over the 4 lane groups of a 16-lane row (lanes i, i + 16, i + 32, i + 48): v_permlane16_swap / v_permlane32_swap of two copies leave every lane with both halves of
// a pair -- VALU only, bit-identical to v + __shfl_xor(v, 16) then + __shfl_xor(., 32) (two ds_bpermute round trips through the LDS pipe)
__device__ __forceinline__ float lmv_xsum4(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float lmv_xmax4(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
