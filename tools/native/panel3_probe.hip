// panel3_probe.hip -- third iteration (see panel2_probe.hip);
// panel2_probe.hip -- second iteration of the panel GEMM experiment (see panel_probe.hip for the first, with its ablations:
// 8-byte / 32-byte-segment stores cost 22 us of 57, and the barrier -> ds_read -> MFMA lock-step of all 8 waves left the matrix
// pipe idle for the LDS latency of every slot).
//   * A panel (<= 128 rows x K) lives in REGISTERS as MFMA fragments (32 rows per wave, loaded once);
//   * W streams through a ring of NS 16-KB slots [128 n][64 k] (LDS-DMA, counted vmcnt, one raw barrier per slot);
//   * the B fragments of slot s + 1 are read into a second register set while the MFMAs of slot s issue;
//   * the epilogue of tile t (bias, GELU, bf16, per-wave LDS transpose, 16-byte row-contiguous stores) is deferred into the
//     slots of tile t + 1, in the same basic block as their MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 panel2_probe.hip -o panel2_probe     run: ./panel2_probe [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}

struct PArgs {
  const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* C;
  int M, N, K;
  int nsplit;      // column ranges per row panel (grid = panels * nsplit)
  const float* gamma; const float* beta; float eps;      // LayerNorm prologue (LN)
  bf16_t* Cpre;    // optional second output: the pre-activation (bias added, before GELU)
};

constexpr int SLOT = 16384;
constexpr int STAGE = 4096;      // per-wave epilogue staging: 32 rows x 64 cols bf16

__device__ __forceinline__ void issue_slot(unsigned char* dst, const bf16_t* __restrict__ W, int N, int K, int n0, int ks, int lane, int wave) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // LDS row r of the slot holds W row n(r): inside every 32-row group the rows are interleaved so that the two MFMA tiles of a
    // pair leave each lane with EIGHT consecutive output columns (one 16-byte store, no LDS transpose in the epilogue):
    // slot row 16 h + 4 g + j  <->  column 8 g + 4 h + j
    const int sg = wave + i * 8, r = sg * 8 + (lane >> 3), p = lane & 7, kc = p ^ ((r >> 1) & 7);
    const int rl = r & 31, nl = (((rl >> 2) & 3) << 3) | ((rl >> 4) << 2) | (rl & 3);
    const bf16_t* src = W + (int64_t)min(n0 + (r & ~31) + nl, N - 1) * K + ks * 64 + kc * 8;
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(dst + sg * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8_t bfrag(const unsigned char* sb, int r, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(sb + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
}

template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int N_> __device__ __forceinline__ void wait_vm() {
  static_assert(N_ >= 0 && N_ <= 40, "");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// LDS accesses of the epilogue go through inline asm: hipcc orders every LDS access it can see behind the pending LDS-DMA
// writes of the ring with s_waitcnt vmcnt(0)
__device__ __forceinline__ float4 lds_read16f(unsigned a) { float4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ uint4 lds_read16u(unsigned a) { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ void lds_write8(unsigned a, uint2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float ex = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return 0.5f * x * (1.0f + copysignf(1.0f - poly * ex, x));
}

// epilogue of one finished tile, in two halves (ti = 0 / 1: 16 rows x 64 columns each): bias (+ pre-activation copy) (+ GELU), bf16,
// and 16-byte stores straight from the accumulators (the interleaved W rows of issue_slot make a tile pair 8 consecutive columns)
template <bool GELU, bool PRE>
__device__ __forceinline__ void epilogue_half(const f32x4_t (&acc)[2][4], int ti, const PArgs& g, unsigned sBias, unsigned sStage, int r0, int R, int n_tile0,
                                              int nt0, int wm, int wn, int lane) {
  const int i = lane & 15, gq = lane >> 4;
  const int ml = wm * 32 + ti * 16 + i;
  float4 b[4];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const unsigned ba = sBias + (n_tile0 + wn * 64 + p * 32 + gq * 8) * 4;
    b[2 * p] = lds_read16f(ba); b[2 * p + 1] = lds_read16f(ba + 16);
  }
  lds_wait();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f32x4_t a0 = acc[ti][2 * p], a1 = acc[ti][2 * p + 1];
    float v[8] = {a0[0] + b[2 * p].x, a0[1] + b[2 * p].y, a0[2] + b[2 * p].z, a0[3] + b[2 * p].w,
                  a1[0] + b[2 * p + 1].x, a1[1] + b[2 * p + 1].y, a1[2] + b[2 * p + 1].z, a1[3] + b[2 * p + 1].w};
    const int n = nt0 * 128 + n_tile0 + wn * 64 + p * 32 + gq * 8;
    const bool ok = ml < R && n < g.N;
    const int64_t o = (int64_t)(r0 + ml) * g.N + n;
    if (PRE) { if (ok) *reinterpret_cast<uint4*>(g.Cpre + o) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])); }
    if (GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
    }
    if (ok) *reinterpret_cast<uint4*>(g.C + o) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
  }
}

// stores of the deferred epilogue sit in the same in-order vmcnt queue as the ring's LDS-DMA: a wait that must leave the (NS - 2)
// youngest slots in flight has to leave the stores issued in the last NS - 1 iterations in flight as well
template <int KS, int NS, int SPS> constexpr int extra_stores(int ks, int e0) {
  int n = 0;
  for (int j = 0; j <= NS - 2; ++j) { const int k = ks - j; if (k == e0 || k == e0 + 1) n += SPS; }
  return n;
}

template <int K, int NS, bool GELU, int ABL, bool LN = false, bool PRE = false>
__global__ __launch_bounds__(512, 2) void panel2_fwd(const PArgs g) {
  constexpr int KS = K / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sB = smem;
  const unsigned sStageBase = (unsigned)(uintptr_t)(smem + NS * SLOT);
  const unsigned sBias = sStageBase + 8 * STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned sStage = sStageBase + wave * STAGE;
  // rows / columns of this workgroup
  const int npanels = gridDim.x / g.nsplit, pnl = blockIdx.x / g.nsplit, part = blockIdx.x % g.nsplit;
  const int T16 = (g.M + 15) >> 4, base = T16 / npanels, rem = T16 % npanels;
  const int start = pnl * base + min(pnl, rem), cnt = base + (pnl < rem ? 1 : 0);
  const int r0 = start * 16, R = min(cnt * 16, g.M - start * 16);
  const int NT_all = (g.N + 127) >> 7, tb = NT_all / g.nsplit, tr = NT_all % g.nsplit;
  const int nt0 = part * tb + min(part, tr), NT = tb + (part < tr ? 1 : 0);
  if (R <= 0 || NT <= 0) return;
  {
    float* sb = reinterpret_cast<float*>(smem + NS * SLOT + 8 * STAGE);
    for (int i = tid; i < NT * 128; i += 512) sb[i] = (nt0 * 128 + i < g.N) ? g.bias[nt0 * 128 + i] : 0.f;
  }
  bf16x8_t af[2][KS * 2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int row = min(wm * 32 + ti * 16 + (lane & 15), R - 1);
#pragma unroll
    for (int kk = 0; kk < KS * 2; ++kk)
      af[ti][kk] = *reinterpret_cast<const bf16x8_t*>(g.A + (int64_t)(r0 + row) * K + kk * 32 + (lane >> 4) * 8);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if constexpr (LN) {
    // LayerNorm prologue IN REGISTERS: a row lives in the 4 lanes (lane & 15) + 16 j, two-pass fp32 statistics
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      float sum = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS * 2; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += (float)af[ti][kk][e];
      sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / K);
      float var = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS * 2; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)af[ti][kk][e] - mean; var = fmaf(d, d, var); }
      var += __shfl_xor(var, 16, 64); var += __shfl_xor(var, 32, 64);
      const float rstd = rsqrtf(var * (1.0f / K) + g.eps);
#pragma unroll
      for (int kk = 0; kk < KS * 2; ++kk) {
        const int k0 = kk * 32 + (lane >> 4) * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(g.gamma + k0), g1 = *reinterpret_cast<const float4*>(g.gamma + k0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(g.beta + k0), b1 = *reinterpret_cast<const float4*>(g.beta + k0 + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        bf16x8_t y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (__bf16)(((float)af[ti][kk][e] - mean) * rstd * gm[e] + bt[e]);
        af[ti][kk] = y;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int kk = 0; kk < KS * 2; ++kk) asm volatile("" : "+v"(af[ti][kk]));

  // logical slot s -> (tile s / KS, k-step s % KS), physical slot s % NS; slots past the end wrap around (harmless re-loads)
  int is_nt = 0, is_ks = 0, is_ph = 0;      // next slot to ISSUE
  auto issue_next = [&]() {
    if constexpr ((ABL & 2) == 0) issue_slot(sB + is_ph * SLOT, g.W, g.N, K, (nt0 + is_nt) * 128, is_ks, lane, wave);
    is_ph = (is_ph + 1 == NS) ? 0 : is_ph + 1;
    if (++is_ks == KS) { is_ks = 0; is_nt = (is_nt + 1 == NT) ? 0 : is_nt + 1; }
  };
#pragma unroll
  for (int s = 0; s < NS; ++s) issue_next();
  int cur_ph = 0;                            // physical slot being computed
  bf16x8_t bq0[4], bq1[4];                   // B fragments of the half-slot (hh = 0 / 1) in flight
  const int brow = wn * 64 + (lane & 15), bkc = lane >> 4;
  auto read_half = [&](bf16x8_t (&dst)[4], int ph, int hh) {
    const unsigned char* sb = sB + ph * SLOT;
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) dst[tj] = bfrag(sb, brow + tj * 16, hh * 4 + bkc);
  };
  wait_vm<2 * (NS - 1)>();                   // slot 0 has landed (this wave's part) ...
  __builtin_amdgcn_s_barrier();              // ... and everybody else's
  asm volatile("" ::: "memory");
  read_half(bq0, 0, 0);

  f32x4_t acc[2][4], prev[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; prev[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  // the two waves of a SIMD (wave, wave + 4) run their deferred epilogues in DIFFERENT slots, so one of them is MFMA-only meanwhile
  constexpr int e0 = (KS >= 4) ? 1 : 0;          // slots of the deferred epilogue (halves ti = 0, 1 of the previous tile)
  constexpr int SPS = 2 * (1 + (PRE ? 1 : 0));   // stores per epilogue slot and wave

  for (int nt = 0; nt < NT; ++nt) {
    static_for<KS>([&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      // ---- first half: MFMAs of (slot, hh = 0); the fragments of hh = 1 stream in behind them
      read_half(bq1, cur_ph, 1);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          if constexpr ((ABL & 4) == 0) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq0[tj], af[ti][ks * 2], acc[ti][tj], 0, 0, 0);
          else asm volatile("" ::"v"(bq0[tj]), "v"(af[ti][ks * 2]));
        }
      if constexpr ((ABL & 1) == 0) {
        if (nt > 0 && ks == e0) epilogue_half<GELU, PRE>(prev, 0, g, sBias, sStage, r0, R, (nt - 1) * 128, nt0, wm, wn, lane);
        if (nt > 0 && ks == e0 + 1) epilogue_half<GELU, PRE>(prev, 1, g, sBias, sStage, r0, R, (nt - 1) * 128, nt0, wm, wn, lane);
      }
      // every read of this slot is in registers; the next slot has landed for this wave -- and, past the barrier, for all waves
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        constexpr int xs = ((ABL & 1) == 0 && (ABL & 8) == 0) ? extra_stores<KS, NS, SPS>(ks, e0) : 0;
        if (xs > 0 && nt > 0) wait_vm<2 * (NS - 2) + xs>(); else wait_vm<2 * (NS - 2)>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_next();                           // into the slot just vacated
      cur_ph = (cur_ph + 1 == NS) ? 0 : cur_ph + 1;
      // ---- second half: MFMAs of (slot, hh = 1); hh = 0 of the NEXT slot streams in
      read_half(bq0, cur_ph, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          if constexpr ((ABL & 4) == 0) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq1[tj], af[ti][ks * 2 + 1], acc[ti][tj], 0, 0, 0);
          else asm volatile("" ::"v"(bq1[tj]), "v"(af[ti][ks * 2 + 1]));
        }
    });
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { prev[i][j] = acc[i][j]; acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  }
  if constexpr ((ABL & 1) == 0) {
    epilogue_half<GELU, PRE>(prev, 0, g, sBias, sStage, r0, R, (NT - 1) * 128, nt0, wm, wn, lane);
    epilogue_half<GELU, PRE>(prev, 1, g, sBias, sStage, r0, R, (NT - 1) * 128, nt0, wm, wn, lane);
  } else if (g.M < 0) {
    epilogue_half<GELU, PRE>(prev, 0, g, sBias, sStage, r0, R, (NT - 1) * 128, nt0, wm, wn, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static bf16_t f2bf_h(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float bf2f_h(bf16_t v) { unsigned u = ((unsigned)v) << 16; float f; memcpy(&f, &u, 4); return f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Ctx {
  int M, N, K; std::vector<bf16_t> hA, hW; std::vector<float> hb, hg, hbt; bf16_t *dA, *dW, *dC, *dP; float *db, *dg, *dbt; hipEvent_t e0, e1;
};

static double gelu_h(double x) { return 0.5 * x * (1.0 + erf(x / sqrt(2.0))); }

template <typename F>
static void bench(const char* name, Ctx& c, int grid, int lds, bool gelu, bool check, F launch, bool ln = false, bool pre = false) {
  CK(hipMemset(c.dC, 0xff, (size_t)c.M * c.N * 2));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  double maxerr = 0, maxref = 1e-30;
  if (check) {
    std::vector<bf16_t> hC((size_t)c.M * c.N), hP;
    CK(hipMemcpy(hC.data(), c.dC, hC.size() * 2, hipMemcpyDeviceToHost));
    if (pre) { hP.resize(hC.size()); CK(hipMemcpy(hP.data(), c.dP, hP.size() * 2, hipMemcpyDeviceToHost)); }
    std::vector<int> rows = {0, 1, 15, 16, 31, 32, 95, 96, 111, 112, 127, 128, c.M - 1, c.M - 2, c.M - 17, c.M / 2, c.M / 3};
    for (int i = 0; i < 40; ++i) rows.push_back((int)((1103515245u * (unsigned)(i + 7) + 12345u) % (unsigned)c.M));
    for (int m : rows) {
      if (m < 0 || m >= c.M) continue;
      std::vector<double> arow(c.K);
      for (int k = 0; k < c.K; ++k) arow[k] = bf2f_h(c.hA[(size_t)m * c.K + k]);
      if (ln) {
        double mu = 0, var = 0;
        for (int k = 0; k < c.K; ++k) mu += arow[k];
        mu /= c.K;
        for (int k = 0; k < c.K; ++k) var += (arow[k] - mu) * (arow[k] - mu);
        const double rs = 1.0 / sqrt(var / c.K + 1e-6);
        for (int k = 0; k < c.K; ++k) arow[k] = bf2f_h(f2bf_h((float)((arow[k] - mu) * rs * c.hg[k] + c.hbt[k])));
      }
      for (int n = 0; n < c.N; ++n) {
        double s = c.hb[n];
        for (int k = 0; k < c.K; ++k) s += arow[k] * (double)bf2f_h(c.hW[(size_t)n * c.K + k]);
        if (pre) { const double gp = bf2f_h(hP[(size_t)m * c.N + n]); maxerr = fmax(maxerr, fabs(gp - s)); }
        if (gelu) s = gelu_h(s);
        const double got = bf2f_h(hC[(size_t)m * c.N + n]);
        maxerr = fmax(maxerr, fabs(got - s)); maxref = fmax(maxref, fabs(s));
      }
    }
  }
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(c.e0));
  const int it = 30;
  for (int i = 0; i < it; ++i) launch();
  CK(hipEventRecord(c.e1));
  CK(hipEventSynchronize(c.e1));
  float ms; CK(hipEventElapsedTime(&ms, c.e0, c.e1));
  const double us = ms * 1e3 / it, tf = 2.0 * c.M * c.N * c.K / us / 1e6;
  printf("%-40s grid %4d lds %6d : %8.1f us %8.1f TFLOP/s   rel err %.2e %s\n", name, grid, lds, us, tf, maxerr / maxref,
         !check ? "(unchecked ablation)" : maxerr / maxref < 6e-3 ? "ok" : "WRONG");
  fflush(stdout);
}

template <int K, int NS, bool GELU, int ABL, bool LN = false, bool PRE = false>
static void run2(Ctx& c, int rows_per_panel, int nsplit) {
  const int T16 = (c.M + 15) / 16, npanels = (T16 + rows_per_panel / 16 - 1) / (rows_per_panel / 16);
  const int NT = (c.N + 127) / 128;
  if (nsplit > NT) return;
  const int grid = npanels * nsplit;
  const int lds = NS * SLOT + 8 * STAGE + ((NT + nsplit - 1) / nsplit) * 128 * 4;
  auto kern = panel2_fwd<K, NS, GELU, ABL, LN, PRE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PArgs g{c.dA, c.dW, c.db, c.dC, c.M, c.N, c.K, nsplit, c.dg, c.dbt, 1e-6f, c.dP};
  char nm[96]; snprintf(nm, sizeof nm, "P3 K=%d ring=%d rows=%d ns=%d gelu=%d ln=%d pre=%d abl=%d", K, NS, rows_per_panel, nsplit, (int)GELU, (int)LN, (int)PRE, ABL);
  bench(nm, c, grid, lds, GELU, ABL == 0 || ABL == 8, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, g); }, LN, PRE);
}

int main(int argc, char** argv) {
  Ctx c;
  c.M = argc > 1 ? atoi(argv[1]) : 27136; c.N = argc > 2 ? atoi(argv[2]) : 1536; c.K = argc > 3 ? atoi(argv[3]) : 384;
  printf("panel2 probe: M=%d N=%d K=%d bf16\n", c.M, c.N, c.K);
  c.hA.resize((size_t)c.M * c.K); c.hW.resize((size_t)c.N * c.K); c.hb.resize(c.N);
  unsigned st = 12345u;
  auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : c.hA) v = f2bf_h(rnd());
  for (auto& v : c.hW) v = f2bf_h(rnd() * 0.05f);
  for (auto& v : c.hb) v = rnd();
  CK(hipMalloc(&c.dA, c.hA.size() * 2)); CK(hipMalloc(&c.dW, c.hW.size() * 2)); CK(hipMalloc(&c.dC, (size_t)c.M * c.N * 2)); CK(hipMalloc(&c.db, c.N * 4));
  CK(hipMemcpy(c.dA, c.hA.data(), c.hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(c.dW, c.hW.data(), c.hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(c.db, c.hb.data(), c.N * 4, hipMemcpyHostToDevice));
  c.hg.resize(c.K); c.hbt.resize(c.K);
  for (auto& v : c.hg) v = 1.0f + 0.3f * rnd();
  for (auto& v : c.hbt) v = 0.2f * rnd();
  CK(hipMalloc(&c.dg, c.K * 4)); CK(hipMalloc(&c.dbt, c.K * 4)); CK(hipMalloc(&c.dP, (size_t)c.M * c.N * 2));
  CK(hipMemcpy(c.dg, c.hg.data(), c.K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(c.dbt, c.hbt.data(), c.K * 4, hipMemcpyHostToDevice));
  CK(hipEventCreate(&c.e0)); CK(hipEventCreate(&c.e1));
  if (c.K == 384) {
    run2<384, 4, false, 0>(c, 112, 1); run2<384, 4, false, 8>(c, 112, 1); run2<384, 5, false, 0>(c, 112, 1); run2<384, 4, false, 0>(c, 128, 1);
    run2<384, 4, true, 0>(c, 112, 1); run2<384, 4, true, 0, false, true>(c, 112, 1); run2<384, 4, false, 0, true>(c, 112, 1); run2<384, 4, true, 0, true, true>(c, 112, 1);
    run2<384, 4, false, 1>(c, 112, 1);
  } else if (c.K == 192) {
    run2<192, 4, false, 0>(c, 128, 1); run2<192, 6, false, 0>(c, 128, 1); run2<192, 4, true, 0, true, true>(c, 128, 1); run2<192, 4, false, 1>(c, 128, 1);
  } else {
    printf("unsupported K\n");
  }
  return 0;
}
