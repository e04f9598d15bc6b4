// fused.hip -- the fused entry points of the block hot path (`ln_linear`, `mlp_fused`): LayerNorm folded into the Linear
// that consumes it, and the whole MLP half of a LeMeBlock (models/lemevit.py:526-530 behind :563-564 / :633,635) in ONE kernel with
// the 4C-wide hidden activations never leaving the chip.
//
// LayerNorm folding.  With x_hat = (x - mean) * rstd:
//     LN(x) W^T + b = rstd * (x (gamma . W)^T  -  mean * colsum(gamma . W))  +  (b + W beta)
// so the GEMM runs on the RAW token rows (LDS-DMA straight from the residual stream, no normalised copy in HBM) against the folded
// weight W' = bf16(gamma . W), and the LayerNorm becomes a per-row scale / per-column shift of the fp32 accumulators.  colsum is taken
// over the ROUNDED W', so the mean term cancels exactly against what the matrix pipe accumulated.  lmv_ln_fold builds (W', colsum, b').
//
// mlp_fused_kernel<C>: a workgroup of 8 waves owns 128 token rows for the whole MLP:
//     X   the 128 x C raw rows, resident in LDS as C / 64 (C / 32) k-tile images (the panel layout of gemm_tiles.h), loaded ONCE;
//         row statistics (two-pass, fp32) are computed from it by the waves that own the rows;
//     for every 128-wide chunk j of the hidden dimension:
//         GEMM1  acc1[128 x 128]  = X . W1'[j]^T      W1' panels (16 KB, [128 x 64]) streamed through a 2..4-slot LDS-DMA ring (counted vmcnt)
//         LN + bias + exact GELU on the accumulators -> H (128 x 128 bf16, two k-tile images in LDS)
//         GEMM2  acc2[128 x C]   += H . W2[:, j]^T    W2 panels streamed through the same ring, C / 128 output panels per wave
//     epilogue: out = x + row_scale * (acc2 + b2), through the per-wave LDS transpose of gemm_tiles.h (16-byte row-segment stores).
// Only the weights stream (128 flop per streamed byte, twice the stand-alone GEMM's 64), and per token the kernel moves C in + C out
// instead of the 17 C of LayerNorm + fc1 + fc2 (the hidden tensor alone is 8 C of that).  LDS: C = 384 uses all 160 KB (96 + 32 + 32).
#include <stdlib.h>
#include <atomic>
#include "gemm_tiles.h"

namespace {

// ---- lmv_ln_fold ---------------------------------------------------------------------------------------------------------------------
// one wave per output row n: wf[n, :] = round(gamma * W[n, :]), colsum[n] = sum of the ROUNDED row, bf[n] = bias[n] + W[n, :] . beta
template <typename T>
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int N, int K, T* __restrict__ wf, float* __restrict__ colsum,
                                                      float* __restrict__ bf) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float s = 0.f, d = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)n * K + k);
    const float4 gv = *reinterpret_cast<const float4*>(gamma + k);
    const float4 bv = *reinterpret_cast<const float4*>(beta + k);
    float f[4] = {wv.x * gv.x, wv.y * gv.y, wv.z * gv.z, wv.w * gv.w};
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = bf2f(f2bf(f[e]));
    }
    st4(wf + (int64_t)n * K + k, f);
    s += (f[0] + f[1]) + (f[2] + f[3]);
    d += (wv.x * bv.x + wv.y * bv.y) + (wv.z * bv.z + wv.w * bv.w);
  }
  s = wave_sum(s); d = wave_sum(d);
  if (lane == 0) { colsum[n] = s; bf[n] = d + (bias ? bias[n] : 0.f); }
}

// ---- mlp_fused_kernel ------------------------------------------------------------------------------------------------------------------
struct MlpProb { const bf16_t* x; bf16_t* out; const float* row_scale; int rows, rps, tile_begin, pad_; };
struct MlpArgs {
  MlpProb p[2];
  const bf16_t* w1; const float* s1; const float* b1;      // folded fc1: [Hd, C], colsum [Hd], bias [Hd]
  const bf16_t* w2; const float* b2;                       // fc2: [C, Hd], [C]
  int nprob, Hd;
  float eps;
#ifdef LMV_MLP_TIMING
  unsigned long long* dbg;      // [2 waves][steps][5] s_memtime stamps of workgroup 0 (tools/mlp_timeline.py)
#endif
};

template <int C, int TM> struct MlpCfg {
  static_assert(C % 32 == 0 && C >= 64 && C <= 384, "mlp_fused: C must be a multiple of 32 in [64, 384]");
  static_assert(TM == 64 || TM == 128, "mlp_fused: 64 or 128 token rows per workgroup");
  static constexpr int NW = TM / 16;                         // waves: (TM / 32) x 2, each 32 rows x 64 columns of a 128-wide step
  static constexpr int BK1 = (C % 64 == 0) ? 64 : 32;        // k-tile depth of GEMM1 (= of the resident X images)
  static constexpr int KT1 = C / BK1;
  static constexpr int XT = TM * BK1 * 2;                    // bytes of one X k-tile image
  static constexpr int X_BYTES = KT1 * XT;
  static constexpr int HT = TM * 64 * 2;                     // one H k-tile image ([TM x 64])
  static constexpr int SLOT = PANEL * 64 * 2;                // one ring slot: a weight panel [128 x 64] (GEMM1 at BK1 = 32: [128 x 32])
  static constexpr int NP = (C + 127) / 128;                 // output panels of GEMM2
  static constexpr int STEPS = KT1 + 2 * NP;                 // panel steps per hidden chunk
  // 128 rows: one workgroup per CU, all 160 KB; 64 rows: TWO workgroups per CU (80 KB each) -- the X load, the LayerNorm statistics,
  // the GELU passes and the store tail of one workgroup then run under the other's MFMAs (measured: a lone 8-wave workgroup per CU
  // spent 3/4 of a tile's life outside its MFMA steps)
  static constexpr int BUDGET = TM == 128 ? 160 * 1024 : 80 * 1024;
  // folded fc1 colsum | bias of the whole hidden dimension (<= 4 C) in LDS when it fits beside a 2-slot ring (else read from L2)
  static constexpr int SB_WANT = 2 * 4 * C * 4;
  static constexpr bool SB = X_BYTES + 2 * HT + SB_WANT + 2 * SLOT <= BUDGET;
  static constexpr int SB_BYTES = SB ? SB_WANT : 0;
  static constexpr int ROOM = (BUDGET - X_BYTES - 2 * HT - SB_BYTES) / SLOT;
  static constexpr int NSLOT = ROOM > 4 ? 4 : ROOM;          // ring depth: NSLOT - 1 weight panels in flight under every step
  static_assert(NSLOT >= 2, "mlp_fused: no room for the weight ring");
  static constexpr int LDS = X_BYTES + 2 * HT + SB_BYTES + NSLOT * SLOT;      // X | H (2 k-tiles) | colsum, bias | ring
  static constexpr int NI1 = (PANEL * BK1 * 2 / 1024) / NW, NI2 = 16 / NW;    // LDS-DMA instructions per wave: GEMM1 / GEMM2 panel
  static constexpr int NIX = (TM * BK1 * 2 / 1024) / NW;                      // ... and per X k-tile
  static constexpr int nld(int t) { return (t % STEPS) < KT1 ? NI1 : NI2; }
  // requests YOUNGER than the panel of step t + 1 once step t has issued its prefetch (panels of steps t + 2 .. t + NSLOT - 1)
  static constexpr int younger(int t) { int n = 0; for (int k = 2; k <= NSLOT - 1; ++k) n += nld(t + k); return n; }
};

// (Measured and dropped: issuing the next panel's LDS-DMA requests INSIDE the step, behind the fragment reads, instead of up front --
//  C = 384: 101 -> 119 us, C = 192: 128 -> 141 us, C = 96: 179 -> 175 us: a request issued while ds_reads are in flight costs more.)
// (Also measured and dropped: only the upper half of the waves requesting the panels, twice the pieces each, so that the lower wave of
//  every SIMD starts its MFMAs at once -- C = 384: 105 -> 129 us, C = 192: 125 -> 140 us, C = 96: 180 -> 193 us.)
// One panel step: the wave's 32 x 64 output tile += A[32 rows x BK] . B[64 cols x BK]^T, both operands in LDS panel images.
// a0 / a1 (b0 / b1): LDS byte address of this lane's fragment of row (column) tile 0 for the k-half hh = 0 / 1; the other tiles are
// 16 rows = 16 * ROWB bytes further (the swizzles repeat every 16 rows), which rides in the instruction's immediate offset -- so a
// step needs FOUR address registers, not one per fragment (hipcc hoists every distinct asm address operand out of the chunk loop).
template <int OFF> __device__ __forceinline__ bf16x8_t ld_frag(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate offset is 16 bits");
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int BK, bool DUAL>
__device__ __forceinline__ void step_mma(unsigned a0, unsigned a1, unsigned b0, unsigned b1, f32x4_t (&acc)[2][4]) {
  constexpr int T16 = 16 * BK * 2;      // bytes between consecutive 16-row tiles of a panel image
  auto rd = [&](bf16x8_t (&af)[2], bf16x8_t (&bf)[4], unsigned a, unsigned b) {
    bf[0] = ld_frag<0>(b); bf[1] = ld_frag<T16>(b); bf[2] = ld_frag<2 * T16>(b); bf[3] = ld_frag<3 * T16>(b);
    af[0] = ld_frag<0>(a); af[1] = ld_frag<T16>(a);
  };
  f32x4_t accb[2];
  bf16x8_t af0[2], bf0[4];
  rd(af0, bf0, a0, b0);
  if constexpr (BK == 32) {
    mma_half<false, 2, 0>(af0, bf0, acc, accb, false);
  } else if constexpr (DUAL) {
    bf16x8_t af1[2], bf1[4];
    rd(af1, bf1, a1, b1);
    mma_half<false, 2, 6>(af0, bf0, acc, accb, false);
    mma_half<false, 2, 0>(af1, bf1, acc, accb, false);
  } else {
    mma_half<false, 2, 0>(af0, bf0, acc, accb, false);
    rd(af0, bf0, a1, b1);
    mma_half<false, 2, 0>(af0, bf0, acc, accb, false);
  }
}
// LDS accesses of the chunk loop that hipcc must not see: it orders every LDS access it can see behind ALL pending LDS-DMA writes
// (s_waitcnt vmcnt(0)), which would drain the weight ring once per chunk
__device__ __forceinline__ f32x4_t lds_ld_f4(unsigned addr) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_st_b64(unsigned addr, unsigned lo, unsigned hi) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const u32x2_t d = {lo, hi};
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(d) : "memory");
}
// an address the compiler must re-derive inside the loop (one v_add per step) instead of keeping one register per k-tile alive
__device__ __forceinline__ unsigned opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }

#ifndef LMV_DUAL_MAXC
#define LMV_DUAL_MAXC 320      // widest C whose steps keep BOTH k-halves' fragments in registers (second half's reads under the first half's MFMAs)
#endif
template <int C, int TM>
__global__ __launch_bounds__(TM * 4) void mlp_fused_kernel(const MlpArgs g) {
  using K = MlpCfg<C, TM>;
  using CF = Cfg<TM / 32, 2, 2>;                             // waves of 32 x 64 over a TM x 128 step tile
  constexpr int BK1 = K::BK1, KT1 = K::KT1, NP = K::NP, STEPS = K::STEPS, NW = K::NW, NTHR = TM * 4;
  constexpr bool DUAL = C <= LMV_DUAL_MAXC;                  // register budget: 256 per lane at 2 waves per SIMD (C = 384: no room beside the prefetched colsum / bias; measured equal anyway)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NSLOT = K::NSLOT, D = NSLOT - 1;             // ring depth, prefetch distance in steps
  unsigned char* const sX = smem;
  unsigned char* const sH = smem + K::X_BYTES;
  float* const sSB = reinterpret_cast<float*>(sH + 2 * K::HT);      // [Hd colsum | Hd bias] (K::SB)
  unsigned char* const sR = sH + 2 * K::HT + K::SB_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
  const int pi = (g.nprob > 1 && bid >= g.p[1].tile_begin) ? 1 : 0;
  const MlpProb& Q = g.p[pi];
  bid -= Q.tile_begin;
  const int m0 = bid * TM, M = Q.rows, Hd = g.Hd;
  const int nchunks = Hd / 128;

  // LDS-DMA sources of the weight panels: per-lane element offsets inside a panel (computed once) + a wave-uniform base per step.
  // Segment sg = wave + 8 i of a panel image is 1 KiB = 8 (16) rows of 128 (64) bytes; the lane that lands on physical chunk p of
  // row r fetches logical chunk p ^ swz(r) (gemm_tiles.h: panel_dma).
  constexpr int NI1 = K::NI1, NI2 = K::NI2, NIX = K::NIX;
  unsigned o1[NI1], o2[NP][NI2];
#pragma unroll
  for (int i = 0; i < NI1; ++i) {
    constexpr int ROWB = BK1 * 2, CPR = ROWB / 16, RPS = 1024 / ROWB;
    const int r = (wave + i * NW) * RPS + lane / CPR, kc = (lane % CPR) ^ swz_n<ROWB>(r);
    o1[i] = (unsigned)(r * C + kc * 8);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int i = 0; i < NI2; ++i) {
      const int r = (wave + i * NW) * 8 + lane / 8, kc = (lane % 8) ^ swz_n<128>(r);
      o2[p][i] = (unsigned)(min(p * 128 + r, C - 1) * Hd + kc * 8);      // rows past C are clamped, never masked (their outputs are not stored)
    }
  auto issue_step = [&](int j, auto tc, unsigned char* slot) {
    constexpr int t = decltype(tc)::value;
    if constexpr (t < KT1) {
      const bf16_t* base = g.w1 + (int64_t)j * 128 * C + t * BK1;
#pragma unroll
      for (int i = 0; i < NI1; ++i) __builtin_amdgcn_global_load_lds((glb_vp)(base + o1[i]), (lds_vp)(slot + (wave + i * NW) * 1024), 16, 0, 0);
    } else {
      constexpr int kt2 = (t - KT1) / NP, p = (t - KT1) % NP;
      const bf16_t* base = g.w2 + j * 128 + kt2 * 64;
#pragma unroll
      for (int i = 0; i < NI2; ++i) __builtin_amdgcn_global_load_lds((glb_vp)(base + o2[p][i]), (lds_vp)(slot + (wave + i * NW) * 1024), 16, 0, 0);
    }
  };

  // X rows (raw, un-normalised; rows past the matrix are clamped) and the first weight panels
  {
    constexpr int ROWB = BK1 * 2, CPR = ROWB / 16, RPS = 1024 / ROWB;
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      const int sg = wave + i * NW, r = sg * RPS + lane / CPR, kc = (lane % CPR) ^ swz_n<ROWB>(r);
      const bf16_t* src = Q.x + (int64_t)min(m0 + r, M - 1) * C + kc * 8;
#pragma unroll
      for (int kt = 0; kt < KT1; ++kt) __builtin_amdgcn_global_load_lds((glb_vp)(src + kt * BK1), (lds_vp)(sX + kt * K::XT + sg * 1024), 16, 0, 0);
    }
  }
  static_for<D>([&](auto tc) { issue_step(0, tc, sR + decltype(tc)::value * K::SLOT); });
  if constexpr (K::SB) {
    for (int i = tid; i < Hd; i += NTHR) { sSB[i] = g.s1[i]; sSB[Hd + i] = g.b1[i]; }
  }
  wait_vm<0>();
  __syncthreads();

  // LayerNorm statistics of this wave's 32 rows: lane -> row (lane & 15) of row tile t, k-chunks of lane group lane >> 4 (the MFMA
  // fragment pattern, so every chunk of a row is read by exactly one lane); two passes over LDS, fp32
  float mu[2], rs[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float s = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
      for (int hh = 0; hh < BK1 / 32; ++hh) {
        const bf16x8_t v = frag_bf16<false, BK1>(sX + kt * K::XT, wm * 32 + t * 16, lane, hh);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e];
      }
    s = lmv_xsum4(s);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
      for (int hh = 0; hh < BK1 / 32; ++hh) {
        const bf16x8_t v = frag_bf16<false, BK1>(sX + kt * K::XT, wm * 32 + t * 16, lane, hh);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)v[e] - mean; q += d * d; }
      }
    q = lmv_xsum4(q);
    mu[t] = mean; rs[t] = rsqrtf(q * (1.0f / C) + g.eps);
  }

  f32x4_t acc2[NP][2][4];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[p][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses (see step_mma): token rows 32 wm.. of an A image, weight rows 64 wn.. of a ring slot
  const unsigned lX = (unsigned)(uintptr_t)sX, lH = (unsigned)(uintptr_t)sH, lR = (unsigned)(uintptr_t)sR, lSB = (unsigned)(uintptr_t)sSB;
  const unsigned fa1_0 = frag_off<false, BK1>(wm * 32, lane, 0), fa1_1 = BK1 == 64 ? frag_off<false, BK1>(wm * 32, lane, 1) : 0u;
  const unsigned fb1_0 = frag_off<false, BK1>(wn * 64, lane, 0), fb1_1 = BK1 == 64 ? frag_off<false, BK1>(wn * 64, lane, 1) : 0u;
  const unsigned fa2_0 = frag_off<false, 64>(wm * 32, lane, 0), fa2_1 = frag_off<false, 64>(wm * 32, lane, 1);
  const unsigned fb2_0 = frag_off<false, 64>(wn * 64, lane, 0), fb2_1 = frag_off<false, 64>(wn * 64, lane, 1);
#ifdef LMV_MLP_TIMING
  const bool stampw = blockIdx.x == 0 && (wave == 0 || wave == NW - 1) && lane == 0 && g.dbg;
  unsigned long long* const dbg = g.dbg + (wave == 0 ? 0 : 4096);
  int nst = 0;
#define STAMP(k) do { if (stampw && nst < 800) dbg[nst * 5 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
  int cur = 0;                                                // ring slot of the current step
  for (int j = 0; j < nchunks; ++j) {
    f32x4_t acc1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc1[i][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool more = j + 1 < nchunks;                        // wave-uniform: panels of the next chunk exist
    f32x4_t s4[4], b4[4];                                     // folded colsum / bias of this lane's 16 hidden columns of the chunk
    static_for<STEPS>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      STAMP(0);
      if constexpr (t == KT1 - 1 && !K::SB) {
        // no room for them in LDS (C = 384): requested from L2 one whole step before the GELU pass needs them -- at the point of use
        // the round trip (~2k cycles under load, s_memtime timeline) stood exposed in every chunk: 17 % of the kernel
        const int n0 = j * 128 + wn * 64 + (lane >> 4) * 4;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          const float4 sv = *reinterpret_cast<const float4*>(g.s1 + n0 + tj * 16), bv = *reinterpret_cast<const float4*>(g.b1 + n0 + tj * 16);
          s4[tj] = f32x4_t{sv.x, sv.y, sv.z, sv.w}; b4[tj] = f32x4_t{bv.x, bv.y, bv.z, bv.w};
        }
      }
      // the panel of step t + D goes into the slot everybody left at the barrier that closed the previous step
      {
        const int nxt = cur == 0 ? NSLOT - 1 : cur - 1;
        if constexpr (t + D < STEPS) issue_step(j, std::integral_constant<int, t + D>{}, sR + nxt * K::SLOT);
        else if (more) issue_step(j + 1, std::integral_constant<int, t + D - STEPS>{}, sR + nxt * K::SLOT);
      }
      STAMP(1);
      const unsigned slot = opaque(lR + cur * K::SLOT);
      if constexpr (t < KT1) {
        const unsigned xa = opaque(lX) + t * K::XT;
        step_mma<BK1, DUAL>(xa + fa1_0, xa + fa1_1, slot + fb1_0, slot + fb1_1, acc1);
      } else {
        constexpr int kt2 = (t - KT1) / NP, p = (t - KT1) % NP;
        const unsigned ha = opaque(lH) + kt2 * K::HT;
        step_mma<64, DUAL>(ha + fa2_0, ha + fa2_1, slot + fb2_0, slot + fb2_1, acc2[p]);
      }
      // the panel of step t + 1 has landed (this wave's pieces: the requests of steps t + 2 .. t + D stay in flight; loads retire in
      // order) -- and, past the barrier, everybody's; in the last chunk the ring runs dry, so drain instead of counting
      STAMP(2);
      if (more || t + D < STEPS) wait_vm<K::younger(t)>(); else wait_vm<0>();
      STAMP(3);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      STAMP(4);
#ifdef LMV_MLP_TIMING
      ++nst;
#endif
      cur = cur + 1 == NSLOT ? 0 : cur + 1;
      if constexpr (t == KT1 - 1) {
        // LayerNorm (folded) + bias + exact GELU on the fc1 accumulators -> H.  This wave's 64 hidden columns are exactly k-tile `wn` of H.
        if constexpr (K::SB) {
          const int n0 = j * 128 + wn * 64 + (lane >> 4) * 4;
          const unsigned a = lSB + n0 * 4;
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) { s4[tj] = lds_ld_f4(a + tj * 64); b4[tj] = lds_ld_f4(a + Hd * 4 + tj * 64); }
          wait_lgkm<0>();
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) { asm volatile("" : "+v"(s4[tj])); asm volatile("" : "+v"(b4[tj])); }
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            const int r = wm * 32 + ti * 16 + (lane & 15);
            const float a = rs[ti], ms = mu[ti];
            const f32x4_t v = acc1[ti][tj];
            // u = rstd (acc - mean colsum) + bias' = rstd acc + (bias' - rstd mean colsum), two columns per packed instruction
            const f32x2_t a2 = {a, a}, nm = {-a * ms, -a * ms};
            const f32x2_t u01 = __builtin_elementwise_fma(a2, f32x2_t{v[0], v[1]}, __builtin_elementwise_fma(nm, f32x2_t{s4[tj][0], s4[tj][1]}, f32x2_t{b4[tj][0], b4[tj][1]}));
            const f32x2_t u23 = __builtin_elementwise_fma(a2, f32x2_t{v[2], v[3]}, __builtin_elementwise_fma(nm, f32x2_t{s4[tj][2], s4[tj][3]}, f32x2_t{b4[tj][2], b4[tj][3]}));
            const f32x2_t h01 = gelu_poly2(u01), h23 = gelu_poly2(u23);
            const int kc = tj * 2 + (lane >> 5);
            lds_st_b64(lH + wn * K::HT + r * 128 + ((kc ^ swz_n<128>(r)) << 4) + ((lane >> 4) & 1) * 8, pack_bf2(h01[0], h01[1]), pack_bf2(h23[0], h23[1]));
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (not __syncthreads: hipcc would drain the weight panels in flight with it)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    });
  }

  // out = x + row_scale * (acc2 + b2): per-wave LDS transpose (scratch: the H / ring images, dead now), 16-byte row-segment stores.
  // The residual x is still in LDS -- the raw rows the kernel loaded once -- so it is NOT read from HBM a second time: the kernel's
  // HBM traffic is C in + C out per token.
  static_assert(K::LDS - K::X_BYTES >= NW * 8192, "mlp_fused: epilogue scratch does not fit behind the X images");
  Problem P{};
  P.bias = g.b2; P.res = Q.x; P.row_scale = Q.row_scale; P.out = Q.out; P.M = M; P.rps = Q.rps > 0 ? Q.rps : 1;
  Problem Pn = P;
  Pn.res = nullptr;                                          // prefetch: bias and DropPath scales only
  Epi<bf16_t, CF, 8192> epi;
  static_assert(sizeof(epi.pf) == 4 * sizeof(uint4), "one pass of 32 rows x 8 chunks per wave");
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    epi.prefetch(Pn, LMV_ACT_NONE, C, C, m0, p * 128, wm, wn, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      constexpr int ROWB = BK1 * 2;
      const int cidx = lane + i * 64, r = wm * 32 + (cidx >> 3), n = min(p * 128 + wn * 64 + (cidx & 7) * 8, C - 8);
      const int kt = n / BK1, kc = (n % BK1) >> 3;
      epi.pf[0][i] = *reinterpret_cast<const uint4*>(sX + kt * K::XT + r * ROWB + ((kc ^ swz_n<ROWB>(r)) << 4));
    }
    epi.store(sH, acc2[p], P, LMV_ACT_NONE, C, C, m0, p * 128, wm, wn, lane, wave);
  }
}

template <int C, int TM>
int launch_mlp(const MlpArgs& g, hipStream_t st) {
  constexpr int lds = MlpCfg<C, TM>::LDS;
  auto kern = mlp_fused_kernel<C, TM>;
  static std::atomic<unsigned long long> attr_done{0};      // > 64 KiB of dynamic LDS: opt in once per kernel and device (idempotent)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      LMV_FAIL(LMV_ERR_LAUNCH, "mlp_fused: cannot reserve %d bytes of LDS", lds);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  MlpArgs a = g;
  int tiles = 0;
  for (int i = 0; i < a.nprob; ++i) { a.p[i].tile_begin = tiles; tiles += (a.p[i].rows + TM - 1) / TM; }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(TM * 4), lds, st, a);
  return LMV_OK;
}
// rows per workgroup: 64 (two workgroups per CU) where its LDS image fits 80 KB, else 128 (LMV_MLP_TM = 64 / 128 forces one: A/B runs)
template <int C>
int launch_mlp_c(const MlpArgs& g, hipStream_t st) {
  const int force = lmv_config().mlp_tm;     // (the parity tests run both tile heights in one process through lmv_config_set)
  constexpr bool fits64 = C <= 192;
  if constexpr (fits64) {
    if (force != 128) return launch_mlp<C, 64>(g, st);
  }
  return launch_mlp<C, 128>(g, st);
}

}  // namespace

extern "C" int lmv_ln_fold(const float* w, const float* bias, const float* gamma, const float* beta, int N, int K, void* wf, float* colsum, float* bf,
                           int dtype, void* stream) {
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "ln_fold: unsupported dtype %d", dtype);
  if (N <= 0 || K <= 0 || (K % 8)) LMV_FAIL(LMV_ERR_SHAPE, "ln_fold: N=%d, K=%d (K must be a positive multiple of 8)", N, K);
  if (!w || !gamma || !beta || !wf || !colsum || !bf) LMV_FAIL(LMV_ERR_SHAPE, "ln_fold: null operand");
  if (!lmv_aligned16(w) || !lmv_aligned16(gamma) || !lmv_aligned16(beta) || !lmv_aligned16(wf)) LMV_FAIL(LMV_ERR_SHAPE, "ln_fold: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((N + 3) / 4), block(256);
  if (dtype == LMV_BF16) hipLaunchKernelGGL(ln_fold_kernel<bf16_t>, grid, block, 0, st, w, bias, gamma, beta, N, K, (bf16_t*)wf, colsum, bf);
  else hipLaunchKernelGGL(ln_fold_kernel<float>, grid, block, 0, st, w, bias, gamma, beta, N, K, (float*)wf, colsum, bf);
  LMV_CHECK_LAUNCH("ln_fold");
  return LMV_OK;
}

// Test hook: the activation the fused MLP kernel applies to its hidden values (gelu_poly2, common.h), on its own -- so that the bound on
// |gelu_poly2(u) - GELU_erf(u)| can be tested without a residual / second GEMM hiding it (tests/test_fused_gpu.py::test_gelu_poly_bound).
namespace {
__global__ __launch_bounds__(256) void gelu_poly_eval_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 < n) {
    const f32x2_t v = gelu_poly2(f32x2_t{x[i], x[i + 1]});
    y[i] = v[0]; y[i + 1] = v[1];
  } else if (i < n) {
    y[i] = gelu_poly2(f32x2_t{x[i], 0.f})[0];
  }
}
}  // namespace
extern "C" int lmv_gelu_poly_eval(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) LMV_FAIL(LMV_ERR_SHAPE, "gelu_poly_eval: null operand or n <= 0");
  hipLaunchKernelGGL(gelu_poly_eval_kernel, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  LMV_CHECK_LAUNCH("gelu_poly_eval");
  return LMV_OK;
}

extern "C" int lmv_mlp_fused_supported(int C, int hidden, int dtype) {
  if (dtype != LMV_BF16) return 0;
  if (hidden <= 0 || (hidden % 128) || hidden > 4 * C) return 0;      // (the LDS copy of colsum | bias is sized for the reference's mlp_ratio 4)
  return C == 64 || C == 96 || C == 128 || C == 192 || C == 256 || C == 320 || C == 384;
}

extern "C" int lmv_mlp_fused_fwd(const lmv_mlp_problem* p, int nproblems, const lmv_mlp_weights* w, int C, int hidden, float eps, int dtype, void* stream) {
  if (dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "mlp_fused: bf16 only (the exact-fp32 mode runs LayerNorm + two Linear launches)");
  if (nproblems < 1 || nproblems > 2 || !p || !w) LMV_FAIL(LMV_ERR_SHAPE, "mlp_fused: nproblems must be 1 or 2");
  if (!lmv_mlp_fused_supported(C, hidden, dtype)) LMV_FAIL(LMV_ERR_SHAPE, "mlp_fused: unsupported C=%d hidden=%d (C in {64,96,128,192,256,320,384}, hidden %% 128 == 0, hidden <= 4 C)", C, hidden);
  if (!w->w1f || !w->colsum1 || !w->b1f || !w->w2 || !w->b2) LMV_FAIL(LMV_ERR_SHAPE, "mlp_fused: null weight");
  if (!lmv_aligned16(w->w1f) || !lmv_aligned16(w->colsum1) || !lmv_aligned16(w->b1f) || !lmv_aligned16(w->w2) || !lmv_aligned16(w->b2))
    LMV_FAIL(LMV_ERR_SHAPE, "mlp_fused: operands must be 16-byte aligned");
  MlpArgs g{};
  g.w1 = (const bf16_t*)w->w1f; g.s1 = w->colsum1; g.b1 = w->b1f; g.w2 = (const bf16_t*)w->w2; g.b2 = w->b2;
  g.nprob = nproblems; g.Hd = hidden; g.eps = eps;
#ifdef LMV_MLP_TIMING
  { const char* e = getenv("LMV_MLP_DBG_PTR"); g.dbg = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  for (int i = 0; i < nproblems; ++i) {
    const lmv_mlp_problem& q = p[i];
    if (q.rows <= 0 || q.rows > 0x7fffffffLL / 4) LMV_FAIL(LMV_ERR_SHAPE, "mlp_fused: bad rows %lld", (long long)q.rows);
    if (!q.x || !q.out || !lmv_aligned16(q.x) || !lmv_aligned16(q.out)) LMV_FAIL(LMV_ERR_SHAPE, "mlp_fused: x / out must be non-null and 16-byte aligned");
    if (q.row_scale && q.rows_per_sample <= 0) LMV_FAIL(LMV_ERR_SHAPE, "mlp_fused: rows_per_sample must be > 0 with row_scale");
    MlpProb& P = g.p[i];
    P.x = (const bf16_t*)q.x; P.out = (bf16_t*)q.out; P.row_scale = q.row_scale; P.rows = (int)q.rows; P.rps = q.rows_per_sample;
  }
  hipStream_t st = (hipStream_t)stream;
  if (lmv_mlp_rw96_eligible(p, nproblems, C, hidden)) return lmv_mlp_rw96_fwd(p, nproblems, w, eps, st);      // both weight matrices resident in LDS (csrc/rwmlp.hip)
  int rc;
  switch (C) {
    case 64:  rc = launch_mlp_c<64>(g, st); break;
    case 96:  rc = launch_mlp_c<96>(g, st); break;
    case 128: rc = launch_mlp_c<128>(g, st); break;
    case 192: rc = launch_mlp_c<192>(g, st); break;
    case 256: rc = launch_mlp_c<256>(g, st); break;
    case 320: rc = launch_mlp_c<320>(g, st); break;
    default:  rc = launch_mlp_c<384>(g, st); break;
  }
  if (rc) return rc;
  LMV_CHECK_LAUNCH("mlp_fused");
  return LMV_OK;
}
