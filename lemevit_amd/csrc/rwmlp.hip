// rwmlp.hip -- the MLP half of a C = 96 block with BOTH weight matrices resident in LDS (round 3; inference form of lmv_mlp_fused_fwd):
//
//   out[r, :] = x[r, :] + s[r] * ( GELU( LN(x[r, :]) W1^T + b1 ) W2^T + b2 )          C = 96, hidden = 384, bf16
//
// Which launches: the four stage-1 "D" blocks (403 456 rows at batch 128) and the meta-token rows of the two stage-0 "C" blocks
// (models/lemevit.py:526-530 behind :563-564).  The tile-streaming form (csrc/fused.hip) moves the 148 KB of fc1 + fc2 weights from L2
// into LDS once per 64-row tile -- 0.93 GB per launch for 77 MB of token rows -- and runs at 170 us, bound by that stream.  Here
//   * fc1 (LayerNorm-folded, [384][96]) and fc2 ([96][384]) sit in LDS for the life of the one 8-wave workgroup of a CU (2 x 73.7 KB;
//     fc1 rows of 192 B with the chunk swizzle of csrc/rswgemm.hip, fc2 rows of 768 B with the 16-byte chunks of every 256-byte window
//     XOR-ed by (((c >> 3) & 3) << 2) | (c & 3): both conflict-free for the lane groups of ds_read_b128 under the interleaved row map);
//   * every wave streams 32-row panels through registers: the raw rows are MFMA B-operands, the row statistics come from those registers
//     (two-pass, fp32) and the folded LayerNorm is applied to the fc1 accumulators (rstd (acc - mean colsum) + b', as csrc/fused.hip);
//   * GEMM1 is computed transposed, D[hidden][row], with the hidden rows of a 32-wide chunk dealt to the two MFMA tiles by rsw_nmap: the
//     8 activations a lane ends up with are hidden indices 8 g .. 8 g + 7 of the chunk -- exactly the k-slots of that lane in the B operand
//     of GEMM2 (the same trick as P in the attention kernels, csrc/attn_mfma.hip): the 4C hidden tensor never leaves the registers, not even
//     for LDS; fc2's A fragments are plain 16-byte reads of its rows;
//   * the residual comes from the raw-row registers (their k-slots are the 8 output columns of the lane), the output leaves as 16-byte stores.
// Per 32-row panel and wave: 144 ds_read_b128 (147 KB) and 288 MFMAs -- LDS-read and MFMA time balance at ~9.2 k cycles per 256 rows of a CU.
// Measured on the stage-1 shape (tools/rw_probe.py, 401 408 rows): 125 us against 171 - 190 us for the tile-streaming form.  Ablations of an
// earlier 141 us version: without the GELU polynomial 96, without the GEMM2 MFMAs 123, without re-reading fc1's fragments 130, ONE hidden chunk instead of
// twelve 52 (rows in, statistics, rows out, the weight images: the HBM floor of 154 MB is ~35 us) -- the kernel is bound by VALU issue: a
// wave-chunk costs ~1.6 k cycles of SIMD time of which ~730 are the 8 packed degree-7 evaluations (v_pk_fma_f32 issues at half rate).
#include <atomic>
#include "common.h"

namespace {

struct RwProb { const bf16_t* x; bf16_t* out; const float* row_scale; int M, rps, wg_begin, nwg; };
struct RwArgs {
  RwProb p[2];
  const bf16_t* w1; const float* s1; const float* b1; const bf16_t* w2; const float* b2;
  int nwg0; float eps;
};

constexpr int RW_C = 96, RW_H = 384, RW_W1ROW = 192, RW_W2ROW = 768;
constexpr int RW_W1_BYTES = RW_H * RW_W1ROW, RW_W2_BYTES = RW_C * RW_W2ROW;
constexpr int RW_LDS = RW_W1_BYTES + RW_W2_BYTES + (2 * RW_H + RW_C) * (int)sizeof(float);

__device__ __forceinline__ int rw_nmap(int t, int i) { return 8 * (i >> 2) + 4 * t + (i & 3); }
__device__ __forceinline__ int rw_swz1(int n) { return (n >> 2) & 3; }
__device__ __forceinline__ int rw_swz2(int c) { return (((c >> 3) & 3) << 2) | (c & 3); }

__device__ __forceinline__ void rw_unpack8(const bf16x8_t& c, float* f) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t r = __builtin_bit_cast(u32x4_t, c);
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(r[e] << 16); f[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
}
__device__ __forceinline__ bf16x8_t rw_pack8(const float* v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t u = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  return __builtin_bit_cast(bf16x8_t, u);
}

// gelu_poly2 (common.h) on two pairs at once, the two Horner chains interleaved statement by statement: a dependent packed op waits for its
// predecessor, so one chain at a time costs a wait state per step
__device__ __forceinline__ void rw_gelu2x2(f32x2_t& a, f32x2_t& b) {
  f32x2_t sa = a * 0.25f, sb = b * 0.25f;
  sa[0] = __builtin_amdgcn_fmed3f(sa[0], -1.0f, 1.0f); sa[1] = __builtin_amdgcn_fmed3f(sa[1], -1.0f, 1.0f);
  sb[0] = __builtin_amdgcn_fmed3f(sb[0], -1.0f, 1.0f); sb[1] = __builtin_amdgcn_fmed3f(sb[1], -1.0f, 1.0f);
  const f32x2_t ua = sa * sa, ub = sb * sb;
  f32x2_t qa = {-1.6300047636032104f, -1.6300047636032104f}, qb = qa;
#define RW_STEP(c) qa = __builtin_elementwise_fma(qa, ua, f32x2_t{c, c}); qb = __builtin_elementwise_fma(qb, ub, f32x2_t{c, c})
  RW_STEP(7.93373966217041f); RW_STEP(-16.877059936523438f); RW_STEP(20.921268463134766f); RW_STEP(-17.09065055847168f);
  RW_STEP(9.8812894821167f); RW_STEP(-4.233964920043945f); RW_STEP(1.595382571220398f);
#undef RW_STEP
  a = a * __builtin_elementwise_fma(sa, qa, f32x2_t{0.5f, 0.5f});
  b = b * __builtin_elementwise_fma(sb, qb, f32x2_t{0.5f, 0.5f});
}

__global__ __launch_bounds__(512, 1) void rw_mlp96_kernel(const RwArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // fc1 image | fc2 image | colsum [384] | b1' [384] | b2 [96]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool second = (int)blockIdx.x >= g.nwg0;
  RwProb P;
#define RW_SEL(f) P.f = second ? g.p[1].f : g.p[0].f
  RW_SEL(x); RW_SEL(out); RW_SEL(row_scale); RW_SEL(M); RW_SEL(rps); RW_SEL(wg_begin); RW_SEL(nwg);
#undef RW_SEL
  unsigned char* const sw1 = smem;
  unsigned char* const sw2 = smem + RW_W1_BYTES;
  float* const scs = reinterpret_cast<float*>(smem + RW_W1_BYTES + RW_W2_BYTES);
  float* const sb1 = scs + RW_H;
  float* const sb2 = sb1 + RW_H;
  // fc1 image: logical 16-byte chunk c = 4 ks + q of row n at physical chunk 4 ks + (q ^ swz1(n));
  // fc2 image: logical chunk c8 (8 hidden values) of row c at physical chunk (c8 & ~15) | ((c8 & 15) ^ swz2(c)).
  // 9 + 9 chunks per thread, all 18 loads in flight before the first LDS store (one L2 round trip per workgroup, not eighteen)
  {
    uint4 v1[9], v2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v1[i] = *reinterpret_cast<const uint4*>(g.w1 + (long long)(tid + i * 512) * 8);
#pragma unroll
    for (int i = 0; i < 9; ++i) v2[i] = *reinterpret_cast<const uint4*>(g.w2 + (long long)(tid + i * 512) * 8);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int idx = tid + i * 512, n = idx / 12, c = idx - n * 12;
      *reinterpret_cast<uint4*>(sw1 + n * RW_W1ROW + (((c & ~3) | ((c & 3) ^ rw_swz1(n))) << 4)) = v1[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int idx = tid + i * 512, c = idx / 48, c8 = idx - c * 48;
      *reinterpret_cast<uint4*>(sw2 + c * RW_W2ROW + (((c8 & ~15) | ((c8 & 15) ^ rw_swz2(c))) << 4)) = v2[i];
    }
  }
  for (int i = tid; i < RW_H; i += 512) { scs[i] = g.s1[i]; sb1[i] = g.b1[i]; }
  if (tid < RW_C) sb2[tid] = g.b2[tid];
  __syncthreads();

  const int npanels = (P.M + 31) / 32;
  const int gw = ((int)blockIdx.x - P.wg_begin) * 8 + wave, stride = P.nwg * 8;
  const int grp = lane >> 4, li = lane & 15;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned w1off[2], w2off[2];             // fc1: byte offset inside a 32-row chunk (k-step 0); fc2: byte offset of the row inside a 32-row slot + its swizzle
  int w2x[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = rw_nmap(t, li);
    w1off[t] = n * RW_W1ROW + ((grp ^ rw_swz1(n)) << 4);
    w2off[t] = n * RW_W2ROW;
    w2x[t] = rw_swz2(n);
  }

  auto load_panel = [&](int pn, bf16x8_t (&af)[2][3]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = min(pn * 32 + rt * 16 + li, P.M - 1);
      const bf16_t* p = P.x + (long long)row * RW_C + grp * 8;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) af[rt][ks] = *reinterpret_cast<const bf16x8_t*>(p + ks * 32);
    }
  };

  bf16x8_t cur[2][3], nxt[2][3];
  if (gw < npanels) load_panel(gw, cur);
  for (int pn = gw; pn < npanels; pn += stride) {
    float mean[2], rstd[2], rsc[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float v[8], s = 0.f;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        rw_unpack8(cur[rt][ks], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
      }
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      mean[rt] = s * (1.f / RW_C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        rw_unpack8(cur[rt][ks], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - mean[rt]; q = fmaf(d, d, q); }
      }
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      rstd[rt] = rsqrtf(q * (1.f / RW_C) + g.eps);
      const int row = min(pn * 32 + rt * 16 + li, P.M - 1);
      rsc[rt] = P.row_scale ? P.row_scale[row / P.rps] : 1.f;
    }
    f32x4_t acc2[2][3][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc2[rt][s][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Two waves per SIMD cannot hide an LDS round trip per fragment: fc2's six fragments of a chunk are requested by hand BEFORE the folded
    // LayerNorm + GELU of the chunk (~650 cycles of VALU work cover the round trip) and waited for once; fc1's fragments are plain reads the
    // compiler pipelines inside GEMM1 (prefetching them by hand as well cost 24 registers: the kernel spilled, and every scratch reload made
    // the in-order vmcnt wait for the next panel's rows).
    bf16x8_t w2f[3][2];
#pragma unroll 1
    for (int hc = 0; hc < RW_H / 32; ++hc) {
      // the next panel's rows are requested ten chunks ahead -- not before the loop: vmcnt retires in order, and any vector-memory wait the
      // compiler places at the top of the loop would wait for these loads at once
      if (hc == 2 && pn + stride < npanels) load_panel(pn + stride, nxt);
      f32x4_t acc1[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc1[rt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const unsigned char* w1b = sw1 + hc * 32 * RW_W1ROW;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        bf16x8_t wf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) wf[t] = *reinterpret_cast<const bf16x8_t*>(w1b + w1off[t] + ks * 64);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc1[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], cur[rt][ks], acc1[rt][t], 0, 0, 0);
      }
      // folded LayerNorm + bias + GELU on the lane's hidden values 32 hc + 8 grp + {0..7} (tile t: + 4 t + {0..3}); packed as the B operand of GEMM2
      const int h0 = hc * 32 + 8 * grp;
      const float4 c0 = *reinterpret_cast<const float4*>(scs + h0), c1 = *reinterpret_cast<const float4*>(scs + h0 + 4);
      const float4 d0 = *reinterpret_cast<const float4*>(sb1 + h0), d1 = *reinterpret_cast<const float4*>(sb1 + h0 + 4);
      const int c8 = hc * 4 + grp;                   // logical 16-byte chunk of the fc2 rows: hidden values 32 hc + 8 grp .. + 7
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned a = lds0 + RW_W1_BYTES + s * 32 * RW_W2ROW + w2off[t] + (((c8 & ~15) | ((c8 & 15) ^ w2x[t])) << 4);
          asm volatile("ds_read_b128 %0, %1" : "=v"(w2f[s][t]) : "v"(a) : "memory");
        }
      bf16x8_t hf[2];
      {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const float nm = -mean[rt];
          u32x4_t pk;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float4 c4 = t ? c1 : c0, d4 = t ? d1 : d0;
            f32x2_t y0 = f32x2_t{fmaf(rstd[rt], fmaf(nm, c4.x, acc1[rt][t][0]), d4.x), fmaf(rstd[rt], fmaf(nm, c4.y, acc1[rt][t][1]), d4.y)};
            f32x2_t y1 = f32x2_t{fmaf(rstd[rt], fmaf(nm, c4.z, acc1[rt][t][2]), d4.z), fmaf(rstd[rt], fmaf(nm, c4.w, acc1[rt][t][3]), d4.w)};
            rw_gelu2x2(y0, y1);
            pk[2 * t] = pack_bf2(y0[0], y0[1]); pk[2 * t + 1] = pack_bf2(y1[0], y1[1]);
          }
          hf[rt] = __builtin_bit_cast(bf16x8_t, pk);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(w2f[s][t]));
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) asm volatile("" : "+v"(hf[rt]));
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc2[rt][s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[s][t], hf[rt], acc2[rt][s][t], 0, 0, 0);
    }
    // out = x + s (acc2 + b2): the lane's output columns of slot s are 32 s + 8 grp + {0..7} -- the k-slots of its raw-row fragment ks = s
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int c0 = s * 32 + 8 * grp;
      const float4 b0 = *reinterpret_cast<const float4*>(sb2 + c0), b1 = *reinterpret_cast<const float4*>(sb2 + c0 + 4);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = pn * 32 + rt * 16 + li;
        float xr[8];
        rw_unpack8(cur[rt][s], xr);
        const float v[8] = {fmaf(rsc[rt], acc2[rt][s][0][0] + b0.x, xr[0]), fmaf(rsc[rt], acc2[rt][s][0][1] + b0.y, xr[1]),
                            fmaf(rsc[rt], acc2[rt][s][0][2] + b0.z, xr[2]), fmaf(rsc[rt], acc2[rt][s][0][3] + b0.w, xr[3]),
                            fmaf(rsc[rt], acc2[rt][s][1][0] + b1.x, xr[4]), fmaf(rsc[rt], acc2[rt][s][1][1] + b1.y, xr[5]),
                            fmaf(rsc[rt], acc2[rt][s][1][2] + b1.z, xr[6]), fmaf(rsc[rt], acc2[rt][s][1][3] + b1.w, xr[7])};
        if (row < P.M) *reinterpret_cast<bf16x8_t*>(P.out + (long long)row * RW_C + c0) = rw_pack8(v);
      }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) cur[rt][ks] = nxt[rt][ks];
  }
}

}  // namespace

bool lmv_mlp_rw96_eligible(const lmv_mlp_problem* p, int nproblems, int C, int hidden) {
  if (C != RW_C || hidden != RW_H || !lmv_config().mlp_rw96) return false;
  for (int i = 0; i < nproblems; ++i)
    if (p[i].rows <= 0 || p[i].rows > 0x7fffffffLL / 2048) return false;
  return true;
}

int lmv_mlp_rw96_fwd(const lmv_mlp_problem* p, int nproblems, const lmv_mlp_weights* w, float eps, hipStream_t st) {
  RwArgs a{};
  a.w1 = (const bf16_t*)w->w1f; a.s1 = w->colsum1; a.b1 = w->b1f; a.w2 = (const bf16_t*)w->w2; a.b2 = w->b2; a.eps = eps;
  // one 8-wave workgroup per CU holds the weights: at most 256 in all (a 257th would start when the first one ends and run the whole
  // launch's tail alone); a small problem gets as many as it has 8-panel groups, the big one the rest
  int want[2] = {0, 0}, nwg = 0;
  for (int i = 0; i < nproblems; ++i) { const int panels = (int)((p[i].rows + 31) / 32); want[i] = (panels + 7) / 8; }
  if (want[0] + want[1] > 256) {
    const int small = want[0] <= want[1] ? 0 : 1, big = 1 - small;
    if (want[small] > 128) want[small] = 128;
    want[big] = 256 - want[small];
  }
  for (int i = 0; i < nproblems; ++i) {
    RwProb& q = a.p[i];
    q.x = (const bf16_t*)p[i].x; q.out = (bf16_t*)p[i].out; q.row_scale = p[i].row_scale; q.M = (int)p[i].rows;
    q.rps = p[i].rows_per_sample > 0 ? p[i].rows_per_sample : 1;
    q.wg_begin = nwg; q.nwg = want[i];
    nwg += want[i];
  }
  a.nwg0 = a.p[0].nwg;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rw_mlp96_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RW_LDS) != hipSuccess)
      LMV_FAIL(LMV_ERR_LAUNCH, "mlp_fused: cannot reserve LDS");
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(rw_mlp96_kernel, dim3(nwg), dim3(512), RW_LDS, st, a);
  LMV_CHECK_LAUNCH("mlp_fused (resident weights)");
  return LMV_OK;
}
