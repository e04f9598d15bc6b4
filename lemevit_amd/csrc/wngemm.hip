// wngemm.hip -- "whole-width" GEMM for the narrow-output Linear launches (round 3):  N = 384 output columns, any K % 64 == 0, bf16
//
//   out[r, n] = epilogue( sum_k A[r, k] W[n, k] + bias[n] )
//
// Which launches: the 384-wide outputs of the stage-3 blocks -- attention projection and fc2 forward (residual epilogue; the projection
// together with the norm2 behind it: lmv_linear_res_ln_fwd) and, through transposed weight copies (lmv_block_desc.*_wt), the dX of qkv and
// fc1 together with the LayerNorm backward of their input (lmv_linear_dx_ln_bwd): a workgroup here owns whole rows.  On the 128 x 128 tile
// kernels of gemm.hip these are 636-tile launches over 512 (8-wave kernel) or 1024 (4-wave kernel) resident slots: 1.24 rounds cost two
// (tools/quant_probe.py: 510 tiles 32 us, 513 tiles 47 us at K = 1536), every token row is fetched by three n-tiles, and a workgroup
// issues one LDS-DMA instruction per four MFMAs (a 128 x 128 tile needs 64 B / clk of operands at the full MFMA rate -- the CU's whole
// vector-memory request rate).  Here
//   * ONE workgroup of 8 waves owns a 128-row token panel for ALL 384 columns: 212 workgroups for the stage-3 launch of Base at batch
//     128 -- one round, no tile quantisation, every token row read once;
//   * wave tile 64 x 96 (4 x 6 MFMA tiles, 96 accumulator registers): 10 fragment reads per 24 MFMAs, one LDS-DMA instruction per six
//     MFMAs; per 64-deep k-step the weights [384 x 64] stream through two 48 KB LDS stages, the token rows [128 x 64] through four
//     16 KB slots (all 160 KB of the CU), LDS-DMA with the swizzle on the source address;
//   * the epilogue stores 16 bytes per lane straight from the accumulators: the W rows of a 32-column block are dealt to the MFMA tiles
//     (wn_nmap) so that two tiles leave a lane with 8 consecutive output columns -- no LDS transpose pass.
// Vector-memory operations of the main loop are LDS-DMA only, counted by hand (the fragment reads are inline asm: hipcc would order
// every LDS read it can see behind the pending LDS-DMA writes with vmcnt(0)).
#include <atomic>
#include <stdlib.h>
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

__device__ float g_wn_zero[512];       // bias == NULL reads zeros from here (zero-initialised, never written)

struct WnProb {
  const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* C; const bf16_t* res; const float* row_scale;
  int M, rps;
  // WN_LNBWD: LayerNorm input rows, their (mean, rstd), optional second output C * row_scale[sample] (res = the residual-path gradient)
  const bf16_t* lnx; const float* stats; bf16_t* Cs;
  float* stats_out;                       // WN_RES_LN: (mean, rstd) of the LayerNorm over the output rows (Cs = its normalised output), may be NULL
};
struct WnArgs {
  WnProb p[2];
  int K, npanels0, npanels;
  const float* gamma; float* partial;      // WN_LNBWD: LayerNorm weight; per-workgroup (dgamma | dbeta) rows [npanels][768]
  const float* beta; float eps;            // WN_RES_LN: LayerNorm bias and epsilon (gamma above)
#ifdef LMV_WN_TIMING
  unsigned long long* dbg;      // s_memtime stamps of wave 0 of workgroups 0 and npanels / 2 (tools/wn_timeline.py)
#endif
};

enum { WN_BIAS = 0, WN_RES = 1, WN_LNBWD = 2, WN_RES_LN = 3 };
constexpr int WN_BM = 128, WN_BN = 384, WN_BK = 64;
constexpr int WN_A_BYTES = WN_BM * WN_BK * 2, WN_W_BYTES = WN_BN * WN_BK * 2;      // token slot 16 KB, weight stage 48 KB
constexpr int WN_A_OFF = 2 * WN_W_BYTES, WN_LDS = 2 * WN_W_BYTES + 4 * WN_A_BYTES;    // [weights: 2 stages | tokens: 4 slots] = 160 KB

template <int N_> __device__ __forceinline__ void wn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <int N_> __device__ __forceinline__ void wn_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory"); }

// W row (inside a wave's 96-column strip) that MFMA tile t reads at operand index i (lane & 15): lane group g = lane >> 4 ends up with
// columns 32 (t >> 1) + 8 g + 4 (t & 1) + {0..3} -- tiles (2p, 2p + 1) together: 8 consecutive columns = one 16-byte store
__device__ __forceinline__ int wn_nmap(int t, int i) { return 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3); }
// 16-byte chunk swizzle of row n of the W image ([384 rows][128 B]): with wn_nmap the four lane groups of a ds_read_b128 hit distinct
// bank quads (the pattern of rsgemm.hip, brute-forced there against the lane grouping of MI355X_MICROARCH.md)
__device__ __forceinline__ int wn_swz_w(int n) { return (((n >> 1) & 1) << 1) | (((n >> 3) & 1) << 2); }
__device__ __forceinline__ int wn_swz_a(int r) { return (r >> 1) & 7; }

__device__ __forceinline__ void wn_gstore16(void* p, const f32x4_t& v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ f32x4_t wn_pack8(const float* v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t u = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  return __builtin_bit_cast(f32x4_t, u);
}
__device__ __forceinline__ void wn_unpack8(const f32x4_t& c, float* f) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t r = __builtin_bit_cast(u32x4_t, c);
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(r[e] << 16); f[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void wn_gemm_kernel(const WnArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                   // 2 (rows) x 4 (columns) waves of 64 x 96
#ifdef LMV_WN_TIMING
  const bool tstamp = g.dbg && (threadIdx.x == 0) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2);
  unsigned long long* const tdbg = g.dbg + (blockIdx.x == 0 ? 0 : 128);
  int tstn = 0;
#define WSTAMP() do { if (tstamp && tstn < 120) tdbg[tstn++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP() do { } while (0)
#endif
  WSTAMP();
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int K = g.K, KS = K / WN_BK;

  // the problem this panel belongs to, selected field by field (an indexed kernel-argument read is a dependent scalar load per use)
  const int panel = blockIdx.x;
  const bool second = panel >= g.npanels0;
  WnProb P;
#define WN_SEL(f) P.f = second ? g.p[1].f : g.p[0].f
  WN_SEL(A); WN_SEL(W); WN_SEL(bias); WN_SEL(C); WN_SEL(res); WN_SEL(row_scale); WN_SEL(M); WN_SEL(rps);
  if (EPI == WN_LNBWD) { WN_SEL(lnx); WN_SEL(stats); WN_SEL(Cs); }
  if (EPI == WN_RES_LN) { WN_SEL(Cs); WN_SEL(stats_out); }
#undef WN_SEL
  if (!P.bias) P.bias = g_wn_zero;
  const int m0 = (second ? panel - g.npanels0 : panel) * WN_BM;

  // ---- LDS-DMA.  A k-step is 64 pieces of 1 KiB (8 rows x 128 B): 48 of the weight rows, 16 of the token panel.  The two operands come
  // from different places -- the weights out of L2 (every workgroup streams the same 1.2 MB), the token rows out of HBM (read once) -- and
  // a wave's vector-memory operations retire IN ORDER, so a wave that waits for its weight pieces of the next step also waits for every
  // token piece it requested before them.  The requests are therefore split BY WAVE: waves 0..5 fetch only weight pieces (8 each per
  // k-step, one step ahead, two 48 KB stages), waves 6..7 only token pieces (8 each per k-step, THREE steps ahead, four 16 KB slots) with
  // a counted wait that leaves two steps in flight: 48 KB of token rows per workgroup = 10 MB over the chip, enough to cover the HBM
  // latency at > 3 TB/s.  (In the model the token operand is cold: fc2 reads the 83 MB fc1 has just written next to another 83 MB;
  // with the token pieces one step ahead like the weights the k-step waited for HBM every time: 73 us in situ against 45 us with the
  // operand resident in the MALL -- tools/insitu_gemm.py, tools/cold_probe.py.)
  // The lane that lands on physical chunk pc = lane & 7 of row r fetches logical chunk pc ^ swz(r); sources advance 64 elements per step.
  const bool a_wave = wave >= 6;                             // wave-uniform
  const bf16_t* src[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pc = lane & 7;
    if (a_wave) { const int r = ((wave - 6) + 2 * i) * 8 + (lane >> 3); src[i] = P.A + (long long)min(m0 + r, P.M - 1) * K + ((pc ^ wn_swz_a(r)) << 3); }
    else { const int r = (wave + 6 * i) * 8 + (lane >> 3); src[i] = P.W + (long long)r * K + ((pc ^ wn_swz_w(r)) << 3); }
  }
  // one k-step of this wave's requests: weight stage `wbuf` (0 / 1) or token slot `aslot` (0..3)
  auto issue_step = [&](int wbuf, int aslot) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned char* dst = a_wave ? smem + WN_A_OFF + aslot * WN_A_BYTES + ((wave - 6) + 2 * i) * 1024 : smem + wbuf * WN_W_BYTES + (wave + 6 * i) * 1024;
      __builtin_amdgcn_global_load_lds((glb_vp)src[i], (lds_vp)dst, 16, 0, 0);
      src[i] += WN_BK;
    }
  };

  // ---- fragment read offsets: token rows inside a slot, weight rows inside a stage; k-half 1 = chunk + 4 -> XOR 64 on the byte offset
  unsigned offa[4], offw[6];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = 64 * wm + 16 * t + (lane & 15);
    offa[t] = r * 128 + ((((lane >> 4)) ^ wn_swz_a(r)) << 4);
  }
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int n = 96 * wn + wn_nmap(t, lane & 15);
    offw[t] = n * 128 + ((((lane >> 4)) ^ wn_swz_w(n)) << 4);
  }

  f32x4_t acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // prologue: weights of step 0; token rows of steps 0, 1, 2
  WSTAMP();
  if (a_wave) {
    issue_step(0, 0);
    if (KS > 1) issue_step(0, 1);
    if (KS > 2) issue_step(0, 2);
    if (KS > 2) wn_wait_vm<16>(); else if (KS > 1) wn_wait_vm<8>(); else wn_wait_vm<0>();
  } else {
    issue_step(0, 0);
    wn_wait_vm<0>();
  }
  WSTAMP();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  WSTAMP();

  bf16x8_t af[2][4], wf[2][6];
  auto rd = [&](int set, unsigned wbase, unsigned abase, int h) {      // the 10 fragments of k-half h
#pragma unroll
    for (int t = 0; t < 6; ++t) { const unsigned a = wbase + (offw[t] ^ (h ? 64u : 0u)); asm volatile("ds_read_b128 %0, %1" : "=v"(wf[set][t]) : "v"(a) : "memory"); }
#pragma unroll
    for (int t = 0; t < 4; ++t) { const unsigned a = abase + (offa[t] ^ (h ? 64u : 0u)); asm volatile("ds_read_b128 %0, %1" : "=v"(af[set][t]) : "v"(a) : "memory"); }
  };
  int buf = 0;
  for (int ks = 0; ks < KS; ++ks) {
    const unsigned wbase = lds0 + buf * WN_W_BYTES, abase = lds0 + WN_A_OFF + (ks & 3) * WN_A_BYTES;
    rd(0, wbase, abase, 0);
    rd(1, wbase, abase, 1);
    // this wave's requests up front (stage / slot nobody reads any more: everybody passed the last barrier).  The step is bound by the
    // L2 -> LDS stream, not by the matrix pipe: 64 KB per CU land ~2.5k cycles after their request (s_memtime timeline,
    // tools/wn_timeline.py: ~11 - 14 TB/s out of L2 over the chip) against ~1.6k cycles of fragment reads + MFMAs; spreading the
    // requests between the MFMA groups instead measured 42.2 against 38.2 us on the fc2 shape (27136 x 384 x 1536).
    if (a_wave) { if (ks + 3 < KS) issue_step(0, (ks + 3) & 3); }
    else { if (ks + 1 < KS) issue_step(buf ^ 1, 0); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 0) wn_wait_lgkm<10>(); else wn_wait_lgkm<0>();
#pragma unroll
      for (int t = 0; t < 6; ++t) asm volatile("" : "+v"(wf[h][t]));          // the MFMAs below must not be scheduled above the wait
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(af[h][t]));
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 6; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][tj], af[h][ti], acc[ti][tj], 0, 0, 0);
    }
    WSTAMP();
    // the next step's operands have landed: weight waves drain their queue, token waves leave the steps beyond ks + 1 in flight
    if (a_wave) {
      if (ks + 3 < KS) wn_wait_vm<16>(); else if (ks + 2 < KS) wn_wait_vm<8>(); else wn_wait_vm<0>();
    } else {
      wn_wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WSTAMP();
    buf ^= 1;
  }

  // ---- epilogue: 16-byte stores straight from the accumulators ----------------------------------------------------------------------
  const int cw = 96 * wn + 8 * (lane >> 4);                  // first column of the lane's 8-column group in pair 0
  if constexpr (EPI == WN_LNBWD) {
    // The accumulators hold dy = dY W (the gradient of a LayerNorm's OUTPUT) for 128 whole rows: the LayerNorm backward runs right here
    // (norm.hip::ln_bwd_kernel's formulas on the fp32 dy):  g = dy gamma, x^ = (x - mean) rstd, s1 = mean_c(g), s2 = mean_c(g x^),
    //   dx = rstd (g - s1 - x^ s2) + dres,   dgamma += sum_r dy x^,   dbeta += sum_r dy
    // A row's 384 columns sit in 4 lane groups x 3 column pairs x the 4 waves of a wave row: the row sums meet through LDS (the operand
    // stages are free now); the column sums are reduced over the 16 rows of a lane group by shuffles, over the row tiles in registers and
    // over the two wave rows through LDS -- fixed order, one (dgamma | dbeta) row per workgroup for the caller's partial-row reduce.
    float* const red = reinterpret_cast<float*>(smem);       // [8 waves][64 rows][2] row sums, then [2 wave rows][768] column sums
    f32x4_t gm[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      gm[p][0] = *reinterpret_cast<const f32x4_t*>(g.gamma + cw + 32 * p);
      gm[p][1] = *reinterpret_cast<const f32x4_t*>(g.gamma + cw + 32 * p + 4);
    }
    const float invC = 1.f / (float)WN_BN;
    float mean[4], rstd[4], s1[4], s2[4];
    {
      f32x4_t xr[4][3];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const int rowc = min(m0 + 64 * wm + 16 * ti + (lane & 15), P.M - 1);
        mean[ti] = P.stats[2 * (long long)rowc]; rstd[ti] = P.stats[2 * (long long)rowc + 1];
#pragma unroll
        for (int p = 0; p < 3; ++p) xr[ti][p] = *reinterpret_cast<const f32x4_t*>(P.lnx + (long long)rowc * WN_BN + cw + 32 * p);
      }
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          float xv[8];
          wn_unpack8(xr[ti][p], xv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float gg = acc[ti][2 * p + (e >> 2)][e & 3] * gm[p][e >> 2][e & 3], xh = (xv[e] - mean[ti]) * rstd[ti];
            a1 += gg; a2 = fmaf(gg, xh, a2);
          }
        }
        a1 = lmv_xsum4(a1);
        a2 = lmv_xsum4(a2);
        s1[ti] = a1; s2[ti] = a2;
        if (lane < 16) { red[(wave * 64 + ti * 16 + lane) * 2] = a1; red[(wave * 64 + ti * 16 + lane) * 2 + 1] = a2; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {           // the four waves of this wave row, in a fixed order
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) { const int o = (((wm * 4 + w4) * 64) + ti * 16 + (lane & 15)) * 2; a1 += red[o]; a2 += red[o + 1]; }
      s1[ti] = a1 * invC; s2[ti] = a2 * invC;
    }
    float cg[3][8], cb[3][8];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) { cg[p][e] = 0.f; cb[p][e] = 0.f; }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int row = m0 + 64 * wm + 16 * ti + (lane & 15);
      const int rowc = min(row, P.M - 1);
      const float live = row < P.M ? 1.f : 0.f;
      const float sc = P.row_scale ? P.row_scale[rowc / P.rps] : 1.f;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const long long o = (long long)rowc * WN_BN + cw + 32 * p;
        float xv[8], rv[8], ov[8];
        wn_unpack8(*reinterpret_cast<const f32x4_t*>(P.lnx + o), xv);
        if (P.res) wn_unpack8(*reinterpret_cast<const f32x4_t*>(P.res + o), rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dy = acc[ti][2 * p + (e >> 2)][e & 3] * live, xh = (xv[e] - mean[ti]) * rstd[ti];
          const float gg = dy * gm[p][e >> 2][e & 3];
          ov[e] = rstd[ti] * (gg - s1[ti] - xh * s2[ti]);
          if (P.res) ov[e] += rv[e];
          cg[p][e] = fmaf(dy, xh, cg[p][e]); cb[p][e] += dy;
        }
        if (row < P.M) {
          wn_gstore16(P.C + o, wn_pack8(ov));
          if (P.Cs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] *= sc;
            wn_gstore16(P.Cs + o, wn_pack8(ov));
          }
        }
      }
    }
    // column sums: over the 16 rows of the lane group (shuffles), then over the two wave rows (LDS)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = cg[p][e], b = cb[p][e];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
        cg[p][e] = a; cb[p][e] = b;
      }
    __syncthreads();                             // everybody has read the row sums
    if ((lane & 15) == 0) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[wm * 768 + cw + 32 * p + e] = cg[p][e]; red[wm * 768 + WN_BN + cw + 32 * p + e] = cb[p][e]; }
    }
    __syncthreads();
    float* prow = g.partial + (long long)panel * 768;
    for (int j = tid; j < 768; j += 512) prow[j] = red[j] + red[768 + j];
    return;
  }
  f32x4_t b4[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    b4[p][0] = *reinterpret_cast<const f32x4_t*>(P.bias + cw + 32 * p);
    b4[p][1] = *reinterpret_cast<const f32x4_t*>(P.bias + cw + 32 * p + 4);
  }
  constexpr bool RES = EPI == WN_RES || EPI == WN_RES_LN;
  float rs1[4], rs2[4];                      // WN_RES_LN: sum and sum of squares of the lane's 24 (rounded) outputs of each row
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    const int row = m0 + 64 * wm + 16 * ti + (lane & 15);
    const int rowc = min(row, P.M - 1);
    float sc = 1.f;
    f32x4_t r4[3];
    if (RES) {
      if (P.row_scale) sc = P.row_scale[rowc / P.rps];
#pragma unroll
      for (int p = 0; p < 3; ++p) r4[p] = *reinterpret_cast<const f32x4_t*>(P.res + (long long)rowc * WN_BN + cw + 32 * p);
    }
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[ti][2 * p][e] + b4[p][0][e]; v[4 + e] = acc[ti][2 * p + 1][e] + b4[p][1][e]; }
      if (RES) {
        float r8[8];
        wn_unpack8(r4[p], r8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = r8[e] + sc * v[e];
      }
      const f32x4_t packed = wn_pack8(v);
      if (row < P.M) wn_gstore16(P.C + (long long)row * WN_BN + cw + 32 * p, packed);
      if (EPI == WN_RES_LN) {                  // the LayerNorm below sees what a separate launch would read back: the ROUNDED outputs
        wn_unpack8(packed, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[ti][2 * p][e] = v[e]; acc[ti][2 * p + 1][e] = v[4 + e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) { a1 += v[e]; a2 = fmaf(v[e], v[e], a2); }
      }
    }
    if (EPI == WN_RES_LN) {
      a1 = lmv_xsum4(a1);
      a2 = lmv_xsum4(a2);
      rs1[ti] = a1; rs2[ti] = a2;
    }
  }
  if constexpr (EPI == WN_RES_LN) {
    // LayerNorm over the 128 finished rows (models/lemevit.py:563: norm2 follows the attention half's residual add): the sums of a row's
    // 384 columns meet through LDS across the four waves of the wave row (one-pass fp32 statistics, variance clamped at 0)
    float* const red = reinterpret_cast<float*>(smem);
    if (lane < 16) {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) { red[(wave * 64 + ti * 16 + lane) * 2] = rs1[ti]; red[(wave * 64 + ti * 16 + lane) * 2 + 1] = rs2[ti]; }
    }
    __syncthreads();
    f32x4_t gm[3][2], bt[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        gm[p][q] = *reinterpret_cast<const f32x4_t*>(g.gamma + cw + 32 * p + 4 * q);
        bt[p][q] = *reinterpret_cast<const f32x4_t*>(g.beta + cw + 32 * p + 4 * q);
      }
    const float invC = 1.f / (float)WN_BN;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) { const int o = (((wm * 4 + w4) * 64) + ti * 16 + (lane & 15)) * 2; a1 += red[o]; a2 += red[o + 1]; }
      const float mean = a1 * invC, var = fmaxf(a2 * invC - mean * mean, 0.f), rstd = rsqrtf(var + g.eps);
      const int row = m0 + 64 * wm + 16 * ti + (lane & 15);
      if (row < P.M) {
        if (P.stats_out && wn == 0 && lane < 16) { P.stats_out[2 * (long long)row] = mean; P.stats_out[2 * (long long)row + 1] = rstd; }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (acc[ti][2 * p + (e >> 2)][e & 3] - mean) * rstd * gm[p][e >> 2][e & 3] + bt[p][e >> 2][e & 3];
          wn_gstore16(P.Cs + (long long)row * WN_BN + cw + 32 * p, wn_pack8(v));
        }
      }
    }
  }
  WSTAMP();
#ifdef LMV_WN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WSTAMP();
#endif
}

template <int EPI>
int wn_launch(const WnArgs& a, hipStream_t st) {
  auto kern = wn_gemm_kernel<EPI>;
  constexpr int lds = WN_LDS;
  static std::atomic<unsigned long long> attr_done{0};      // > 64 KiB of dynamic LDS: opt in once per kernel and device (idempotent)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      LMV_FAIL(LMV_ERR_LAUNCH, "wn linear: cannot reserve %d bytes of LDS", lds);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(a.npanels), dim3(512), lds, st, a);
  return LMV_OK;
}

}  // namespace

// Does the whole-width kernel take this forward-form launch?  (bf16; `force`: every shape it can run, for the parity tests)
bool lmv_wn_eligible(const lmv_linear_problem* p, int nproblems, int N, int K, int act, bool force) {
  if (N != WN_BN || (K % WN_BK) || K < WN_BK || act != LMV_ACT_NONE) return false;
  int64_t rows = 0;
  for (int i = 0; i < nproblems; ++i) {
    if (p[i].rows <= 0 || p[i].rows > 0x7fffffffLL / 2048 || p[i].out_pre || p[i].aux) return false;
    if (!p[i].res && p[i].row_scale) return false;
    rows += p[i].rows;
  }
  if (nproblems == 2 && (p[0].res != nullptr) != (p[1].res != nullptr)) return false;      // one kernel instance: one epilogue
  if (force) return true;
  // Measured wins only (tools/quant_probe.py wn, tools/cold_probe.py, profiles/r03_wn_probe.txt).  Forward launches with the residual
  // epilogue (attention projection, fc2): 45.1 -> 39.6 us at K = 1536 with the operands in the MALL, 66 -> 52 us with cold operands.
  // Plain launches (the dX of proj through the transposed weight) stay on the tile kernel: next to the weight-gradient stream a 160 KB
  // workgroup waits for a whole CU to drain, and the step does not gain (profiles/r03_ab_switches.txt).
  if (!p[0].res) return false;
  // ONE round of 128-row panels on the 256 CUs: with fewer than ~128 panels the 128 x 128 tiles fill the chip better (8320 rows: 13 vs 8 us);
  // beyond one round the one-per-CU workgroups lose (102400 rows = 3.1 rounds: 190 vs 178 us at K = 1536; Base at 384^2, B = 64 = 1.2 rounds:
  // train step 49.5 vs 48.1 ms with / without)
  return rows >= 16384 && rows <= 32768;
}

// dX of a Linear through its transposed weight fused with the LayerNorm backward of the Linear's input (include/lemevit_hip.h)
extern "C" int lmv_linear_dx_ln_bwd_supported(int C, int N, int dtype) {
  return dtype == LMV_BF16 && C == WN_BN && N >= WN_BK && (N % WN_BK) == 0;
}
extern "C" size_t lmv_linear_dx_ln_bwd_workspace_bytes(int64_t total_rows, int C) {
  if (total_rows <= 0 || C != WN_BN) return 0;
  return (size_t)((total_rows + WN_BM - 1) / WN_BM + 2) * 2 * C * sizeof(float);
}
extern "C" int lmv_linear_dx_ln_bwd(const lmv_linear_problem* p, const lmv_ln_segment* seg, int nproblems, int C, int N, const float* gamma,
                                    void* workspace, size_t workspace_bytes, int* partial_rows, int dtype, void* stream) {
  if (!lmv_linear_dx_ln_bwd_supported(C, N, dtype)) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx_ln_bwd: C=%d N=%d dtype=%d (bf16, C = 384, N %% 64 == 0)", C, N, dtype);
  if (nproblems < 1 || nproblems > 2 || !p || !seg || !gamma || !workspace || !partial_rows) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx_ln_bwd: null argument / nproblems must be 1 or 2");
  WnArgs a{};
  int npan[2] = {0, 0};
  for (int i = 0; i < nproblems; ++i) {
    const lmv_ln_segment& q = seg[i];
    if (p[i].rows <= 0 || p[i].rows != q.rows || p[i].rows > 0x7fffffffLL / 2048) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx_ln_bwd: bad rows");
    if (!p[i].a || !p[i].w || !q.x || !q.stats || !q.dx || (q.dx_scale && (!q.dx_scaled || q.rows_per_sample <= 0)))
      LMV_FAIL(LMV_ERR_SHAPE, "linear_dx_ln_bwd: null operand");
    if (!lmv_aligned16(p[i].a) || !lmv_aligned16(p[i].w) || !lmv_aligned16(q.x) || !lmv_aligned16(q.dx) || !lmv_aligned16(q.dres) || !lmv_aligned16(q.dx_scaled) || !lmv_aligned16(gamma))
      LMV_FAIL(LMV_ERR_SHAPE, "linear_dx_ln_bwd: operands must be 16-byte aligned");
    WnProb& w = a.p[i];
    w.A = (const bf16_t*)p[i].a; w.W = (const bf16_t*)p[i].w; w.bias = nullptr; w.C = (bf16_t*)q.dx; w.res = (const bf16_t*)q.dres;
    w.row_scale = q.dx_scale; w.M = (int)q.rows; w.rps = q.dx_scale ? (int)q.rows_per_sample : 1;
    w.lnx = (const bf16_t*)q.x; w.stats = q.stats; w.Cs = q.dx_scale ? (bf16_t*)q.dx_scaled : nullptr;
    npan[i] = (int)((q.rows + WN_BM - 1) / WN_BM);
  }
  a.K = N; a.npanels0 = npan[0]; a.npanels = npan[0] + npan[1];
  if (workspace_bytes < (size_t)a.npanels * 768 * sizeof(float) || !lmv_aligned16(workspace))
    LMV_FAIL(LMV_ERR_WORKSPACE, "linear_dx_ln_bwd: workspace %zu < %zu bytes", workspace_bytes, (size_t)a.npanels * 768 * sizeof(float));
  a.gamma = gamma; a.partial = (float*)workspace;
#ifdef LMV_WN_TIMING
  a.dbg = nullptr;
#endif
  double trows_ = 0.;
  for (int i = 0; i < nproblems; ++i) trows_ += (double)p[i].rows;
  LmvTimedLaunch timed(stream, 2.0 * C * N * trows_, 2.0 * trows_ * (N + 3.0 * C) + 2.0 * N * C, LMV_TK_WN_GEMM);          // (dY in; x, dres in, dx out; the transposed weight once)
  if (int rc = wn_launch<WN_LNBWD>(a, (hipStream_t)stream)) return rc;
  LMV_CHECK_LAUNCH("linear_dx_ln_bwd");
  *partial_rows = a.npanels;
  return LMV_OK;
}

// out = res + row_scale (a W^T + bias);  y = LayerNorm(out) -- the attention projection of a block with the norm2 that follows it
extern "C" int lmv_linear_res_ln_fwd_supported(int N, int K, int dtype) { return dtype == LMV_BF16 && N == WN_BN && K >= WN_BK && (K % WN_BK) == 0; }
extern "C" int lmv_linear_res_ln_fwd(const lmv_linear_problem* p, const lmv_ln_segment* seg, int nproblems, int N, int K, const float* gamma, const float* beta,
                                     float eps, int dtype, void* stream) {
  if (!lmv_linear_res_ln_fwd_supported(N, K, dtype)) LMV_FAIL(LMV_ERR_SHAPE, "linear_res_ln_fwd: N=%d K=%d dtype=%d (bf16, N = 384, K %% 64 == 0)", N, K, dtype);
  if (nproblems < 1 || nproblems > 2 || !p || !seg || !gamma || !beta || !(eps > 0.f)) LMV_FAIL(LMV_ERR_SHAPE, "linear_res_ln_fwd: null argument / nproblems must be 1 or 2");
  double trows = 0., tbytes = 2.0 * N * K + 12.0 * N;          // A, residual in; out and LayerNorm(out) out
  if (g_lmv_timing_on)
    for (int i = 0; i < nproblems; ++i) { trows += (double)p[i].rows; tbytes += 2.0 * p[i].rows * (K + 3.0 * N); }
  LmvTimedLaunch timed(stream, 2.0 * N * K * trows, tbytes, LMV_TK_WN_GEMM);
  WnArgs a{};
  int npan[2] = {0, 0};
  for (int i = 0; i < nproblems; ++i) {
    const lmv_ln_segment& q = seg[i];
    if (p[i].rows <= 0 || p[i].rows != q.rows || p[i].rows > 0x7fffffffLL / 2048) LMV_FAIL(LMV_ERR_SHAPE, "linear_res_ln_fwd: bad rows");
    if (!p[i].a || !p[i].w || !p[i].out || !p[i].res || !q.y || p[i].out_pre || p[i].aux || (p[i].row_scale && p[i].rows_per_sample <= 0))
      LMV_FAIL(LMV_ERR_SHAPE, "linear_res_ln_fwd: a, w, out, res and seg.y are required; no out_pre / aux");
    if (!lmv_aligned16(p[i].a) || !lmv_aligned16(p[i].w) || !lmv_aligned16(p[i].out) || !lmv_aligned16(p[i].res) || !lmv_aligned16(q.y) || !lmv_aligned16(p[i].bias) ||
        !lmv_aligned16(gamma) || !lmv_aligned16(beta))
      LMV_FAIL(LMV_ERR_SHAPE, "linear_res_ln_fwd: operands must be 16-byte aligned");
    WnProb& w = a.p[i];
    w.A = (const bf16_t*)p[i].a; w.W = (const bf16_t*)p[i].w; w.bias = p[i].bias; w.C = (bf16_t*)p[i].out; w.res = (const bf16_t*)p[i].res;
    w.row_scale = p[i].row_scale; w.M = (int)p[i].rows; w.rps = p[i].rows_per_sample > 0 ? p[i].rows_per_sample : 1;
    w.Cs = (bf16_t*)q.y; w.stats_out = q.stats;
    npan[i] = (int)((p[i].rows + WN_BM - 1) / WN_BM);
  }
  a.K = K; a.npanels0 = npan[0]; a.npanels = npan[0] + npan[1];
  a.gamma = gamma; a.beta = beta; a.eps = eps;
#ifdef LMV_WN_TIMING
  a.dbg = nullptr;
#endif
  if (int rc = wn_launch<WN_RES_LN>(a, (hipStream_t)stream)) return rc;
  LMV_CHECK_LAUNCH("linear_res_ln_fwd");
  return LMV_OK;
}

int lmv_wn_linear(const lmv_linear_problem* p, int nproblems, int N, int K, int act, hipStream_t st) {
  (void)N; (void)act;
  WnArgs a{};
  int npan[2] = {0, 0};
  for (int i = 0; i < nproblems; ++i) {
    WnProb& q = a.p[i];
    q.A = (const bf16_t*)p[i].a; q.W = (const bf16_t*)p[i].w; q.bias = p[i].bias; q.C = (bf16_t*)p[i].out;
    q.res = (const bf16_t*)p[i].res; q.row_scale = p[i].row_scale;
    q.M = (int)p[i].rows; q.rps = p[i].rows_per_sample > 0 ? p[i].rows_per_sample : 1;
    npan[i] = (int)((p[i].rows + WN_BM - 1) / WN_BM);
  }
  a.K = K; a.npanels0 = npan[0]; a.npanels = npan[0] + npan[1];
#ifdef LMV_WN_TIMING
  { const char* e = getenv("LMV_WN_DBG_PTR"); a.dbg = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  const int rc = p[0].res ? wn_launch<WN_RES>(a, st) : wn_launch<WN_BIAS>(a, st);
  if (rc) return rc;
  LMV_CHECK_LAUNCH("wn linear");
  return LMV_OK;
}
