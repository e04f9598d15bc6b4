// gemm.hip -- MFMA GEMM for the Linear layers of the LeMeViT blocks (gfx950).
//
// One kernel template covers the three contractions a Linear layer needs:
//   fwd  C[r,n] = sum_k  A[r,k]  W[n,k]      (A, W both reduction-contiguous: plain 16-B LDS reads)
//   dx   C[r,k] = sum_n dY[r,n]  W[n,k]      (W is reduction-MAJOR: ds_read_b64_tr_b16 transpose reads)
//   dw   C[n,k] = sum_r dY[r,n]  X[r,k]      (both operands reduction-major; the token reduction is split over
//                                              workgroups that write fp32 partial slabs, summed by a 2nd kernel:
//                                              deterministic, and no same-address atomics, which serialise at
//                                              ~170 ns each across XCDs)
// so the backward pass needs neither transposed weight copies nor transposed activations.
//
// Workgroup tile (template Cfg): 128x128 (4 waves), 256x128 or 256x256 (8 waves); every wave owns 64x64 or 128x64
// outputs = WM x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16; exact-fp32 mode: v_mfma_f32_16x16x4_f32).  With K = 96..512
// a workgroup spends as long in launch + first-load latency + epilogue + store drain as in its k-loop, and only OTHER
// resident workgroups hide that, so occupancy decides: 128x128 tiles over 32-deep k-tiles (32 KB of LDS) under a
// 128-register cap put 4 workgroups on a CU (3 for dW); the larger tiles and 4-wave 64-deep k-tiles (2 per CU) lose on
// every large layer shape of the model (make_plan; measurements in DESIGN.md 4.1).  The forward (and dX launches that cannot
// fill 4 workgroups per CU) runs the same 128x128 tile on EIGHT waves of 32x64 over 64-deep k-tiles (C128w8): still 4 waves
// per SIMD, but every operand row is a whole 128-byte line per k-tile -- the k-loop is bound by the L2 -> LDS operand
// stream, and L2 serves half a line in the slot of a whole one (15 vs 29 TB/s, tools/native/dma_probe.hip).
//
// Operand tiles live in LDS as 128-row (or, transposed, 128-column) PANELS, double-buffered over k-tiles of 32 / 64
// (bf16) or 32 (fp32).  bf16 panels are filled by LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16, so
// the XOR swizzle is applied to the per-lane SOURCE address); fp32 and ragged-K problems use a register-staged
// path (global -> VGPR -> ds_write) on the 128x128 tile.  The swizzles make the 16-byte fragment reads and the
// transpose reads bank-conflict free (derivation in DESIGN.md; SQ_LDS_BANK_CONFLICT = 0 measured).
// Workgroup ids are remapped so that the n-tiles of one m-tile run on the same XCD and share its L2, and so that the four
// workgroups the dispatcher places on one CU (XCD-local ids i, i + 32, i + 64, i + 96) get consecutive tiles (shared A rows
// can meet in the CU's vector L1).
//
// The MFMA operands are passed swapped (W as "A") so every lane ends up with 4 CONSECUTIVE output columns of one
// row.  The epilogue stages the fp32 tile through swizzled LDS, 64 rows at a time, and leaves as whole row
// segments: 16-byte coalesced stores, 16-byte coalesced residual / GELU' operand loads, with bias, GELU,
// DropPath scale and residual add fused.  In dW mode the bias gradient (column sums of dY) rides the matrix pipe:
// one extra MFMA per fragment against an all-ones operand.
#include <stdlib.h>
#include <atomic>
#include <utility>
#include "common.h"

#include "gemm_tiles.h"

namespace {

template <typename T, bool ATR, bool BTR, bool SPLITK, int BK, bool DMA, typename CF, int MINW = 1, int NST = 2, bool LNP = false, bool CONV = false>
__global__ __launch_bounds__(CF::NTHR, MINW) void gemm_kernel(const GemmArgs g) {
  static_assert(!LNP || (!ATR && !BTR && !SPLITK), "the LayerNorm-folded epilogue belongs to the forward contraction");
  static_assert(!CONV || (DMA && ((!ATR && !BTR && !SPLITK) || (ATR && BTR && SPLITK))), "the implicit-GEMM convolution: the LDS-DMA forward and weight-gradient kernels");
  using RS = typename std::conditional<LNP, RowStat<CF::WM>, NoStat>::type;
  RS rstat;
  if constexpr (LNP) rstat.clear();
#ifdef LMV_GEMM_TIMING
  // timeline of wave 0 of the first and of a middle workgroup: [0] entry, [1] first requests issued, [2] first k-tile landed, then per k-tile
  // (request, MFMAs done, barrier passed), then [k-loop end], [epilogue operands requested], [stores issued]
  const bool tstamp = g.dbg && (threadIdx.x == 0) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2);
  unsigned long long* const tdbg = g.dbg + (blockIdx.x == 0 ? 0 : 256);
  int tstn = 0;
#define GSTAMP() do { if (tstamp && tstn < 250) tdbg[tstn++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GSTAMP() do { } while (0)
#endif
  // (Measured and dropped: staggering the first generation of workgroups by a start delay per resident slot, so that the store bursts of
  //  one half of the chip meet the k-loops of the other -- the s_memtime timeline shows every workgroup of a generation storing at once,
  //  each 16-byte store round then takes ~1k cycles = the chip's whole write bandwidth.  Every delay from 4k to 12k cycles only lengthened
  //  the launches by about the delay itself: 5 generations of ~20k-cycle workgroups are too short to win it back.)
  GSTAMP();
  constexpr int PANEL_BYTES = PANEL * BK * (int)sizeof(T);
  constexpr int BUF_BYTES = (CF::PA + CF::PB) * PANEL_BYTES;
  constexpr int WM = CF::WM;
  static_assert(DMA || (CF::PA == 1 && CF::PB == 1 && CF::NTHR == 256), "the register-staged loop is written for the 128x128 tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [NST buffers][A panels | B panels]; reused by the epilogue

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / CF::NWN, wn = wave % CF::NWN;
  // XCD-aware order: hardware places block b on XCD b % 8; give each XCD a contiguous run of logical tiles so
  // the n-tiles of one m-tile share an L2 (speed only -- any placement is correct).
  // dW: the workgroups of one k-split (all tiles of dW read the SAME token rows) sit 8 ids apart -- same XCD, launched
  // together -- so the rows are fetched from HBM once and shared through that XCD's L2.
  int bid, split = 0;
  if constexpr (SPLITK) {
    const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
    if (g.nsplits >= 8) {                // a multiple of 8: XCD x owns splits x, x + 8, ...
      bid = y % g.ntiles;
      split = (y / g.ntiles) * 8 + x;
    } else {                             // 1, 2 or 4: the XCDs that share a split divide its tiles
      split = x % g.nsplits;
      bid = y * (8 / g.nsplits) + x / g.nsplits;
    }
    if (split >= g.nsplits || bid >= g.ntiles) return;
  } else {
    const int T_ = gridDim.x, xcd = blockIdx.x & 7, q = T_ >> 3, r = T_ & 7, len = q + (xcd < r ? 1 : 0);
    int idx = blockIdx.x >> 3;
    // Inside an XCD the dispatcher deals workgroups round-robin over its 32 CUs (tools/native/hwid_probe.hip): the XCD-local
    // workgroups i, i + 32, i + 64, i + 96 share a CU.  Hand those four CONSECUTIVE tiles (same m-tile, neighbouring n-tiles) so
    // that their A rows can meet in the CU's vector L1 instead of each going to L2.
    if (g.cumap) {
      const int chunk = idx & ~127;
      if (chunk + 128 <= len) idx = chunk + ((idx & 31) << 2) + ((idx >> 5) & 3);
    }
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const bool second = g.nprob > 1 && !g.concat && bid >= g.p[1].tile_begin;      // workgroup-uniform
  const int pi = second ? 1 : 0;
  // The problem descriptor is SELECTED field by field instead of indexed (`g.p[pi]`): an indexed kernel-argument read is a scalar load
  // whose address depends on pi, i.e. one dependent s_load + s_waitcnt round trip per use site -- the s_memtime timeline of a stage-3
  // forward launch showed 1.6 - 2.5k cycles between kernel entry and the first LDS-DMA request (of a 22k-cycle workgroup life) and
  // another such chain in front of the epilogue.  With constant addresses all of both problems' fields load in one batch at entry.
  Problem P;
#define LMV_SEL(f) P.f = second ? g.p[1].f : g.p[0].f
  LMV_SEL(a); LMV_SEL(b); LMV_SEL(bias); LMV_SEL(res); LMV_SEL(row_scale); LMV_SEL(aux); LMV_SEL(out); LMV_SEL(out_pre); LMV_SEL(bias_grad);
  LMV_SEL(M); LMV_SEL(Kred); LMV_SEL(rps); LMV_SEL(tiles_m); LMV_SEL(tile_begin);
#undef LMV_SEL
  bid -= P.tile_begin;
  const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
  const int m0 = tm * CF::BM, n0 = tn * CF::BN;
  const int M = P.M, N = g.N;
  // dW of two problems that share dW (x and c rows through the same weights): ONE reduction over the rows of both
  const int kt0 = (P.Kred + BK - 1) / BK;
  const int kt_total = kt0 + ((SPLITK && g.concat) ? (g.p[1].Kred + BK - 1) / BK : 0);
  const int kt_beg = split * g.kt_per_split;
  const int kt_end = min(kt_total, kt_beg + g.kt_per_split);
  if (kt_beg >= kt_end) return;
  // operands of k-tile kt
  // (operand bases and reduction lengths are copied out of the kernel arguments ONCE: addressed through `P` / `g.p[1]` they were
  //  re-read with s_load + s_waitcnt in front of every k-tile's requests)
  const T* const a0 = reinterpret_cast<const T*>(P.a); const T* const b0 = reinterpret_cast<const T*>(P.b);
  const T* const a1 = SPLITK ? reinterpret_cast<const T*>(g.p[1].a) : a0; const T* const b1 = SPLITK ? reinterpret_cast<const T*>(g.p[1].b) : b0;
  const int kred0 = P.Kred, kred1 = SPLITK ? g.p[1].Kred : P.Kred;
  auto src_of = [&](int kt, const T** a, const T** b, int* kred, int* k0) {
    const bool second = SPLITK && kt >= kt0;
    *a = second ? a1 : a0; *b = second ? b1 : b0;
    *kred = second ? kred1 : kred0; *k0 = (second ? kt - kt0 : kt) * BK;
  };

  f32x4_t acc[WM][4], accb[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // this wave's operand panels and offsets inside them
  const int ra0 = wm * WM * 16, cb0 = wn * 64;
  const int pa = ra0 / PANEL, oa = ra0 % PANEL, pb = cb0 / PANEL, ob = cb0 % PANEL;
  const bool do_bsum = SPLITK && (P.bias_grad != nullptr) && tn == 0 && wn == 0;      // wave-uniform

  // epilogue operands (GELU' operand / residual; on the 8-wave kernels also DropPath scales and bias) are requested right after the
  // k-loop, before the first LDS transpose (requesting them before the loop costs the 64-deep kernels their double-buffered fragments)
  constexpr int REGION = NST * BUF_BYTES / CF::NW;      // epilogue: the operand buffers, carved into one private region per wave
  constexpr bool EPI_EARLY = false;
  Epi<T, CF, SPLITK ? 4096 : REGION> epi;      // (unused by the split-K kernels)
  if constexpr (EPI_EARLY) epi.prefetch(P, g.act & 0xff, N, g.ldc, m0, n0, wm, wn, lane);

  int cur = 0;
  if constexpr (DMA) {
    static_assert(sizeof(T) == 2, "the direct-to-LDS path is bf16 only");
    auto issue = [&](unsigned char* buf, int kt) {
      const bf16_t *A16, *B16; int kred, k0;
      src_of(kt, &A16, &B16, &kred, &k0);
#pragma unroll
      for (int q = 0; q < CF::PA; ++q) {
        if constexpr (CONV && !SPLITK) panel_dma_conv<false, BK, CF::NW>(buf + q * PANEL_BYTES, g.cv, M, m0 + q * PANEL, k0, lane, wave);          // forward: the patch rows
        else panel_dma<ATR, BK, CF::NW>(buf + q * PANEL_BYTES, A16, g.lda, M, m0 + q * PANEL, k0, lane, wave);
      }
#pragma unroll
      for (int q = 0; q < CF::PB; ++q) {
        if constexpr (CONV && SPLITK) panel_dma_conv<true, BK, CF::NW>(buf + (CF::PA + q) * PANEL_BYTES, g.cv, N, n0 + q * PANEL, k0, lane, wave);      // weight gradient: the patch columns
        else panel_dma<BTR, BK, CF::NW>(buf + (CF::PA + q) * PANEL_BYTES, B16, g.ldb, N, n0 + q * PANEL, k0, lane, wave);
      }
    };
    if constexpr (CF::NW >= 8 && CF::WM == 2 && NST == 2) {
      // 8-wave kernels (64-deep k-tiles, 1.5x the fragment reads): the plain loop -- hipcc waits for the requested k-tile BEFORE this
      // tile's fragment reads, so LDS-DMA writes and ds_reads never share the LDS; measured 3 % faster here than the overlapped loop below
      issue(smem, kt_beg);
      GSTAMP();
      __syncthreads();                                  // (the compiler drains vmcnt before the barrier: the LDS-DMA landed)
      GSTAMP();
      for (int kt = kt_beg; kt < kt_end; ++kt) {
        if (kt + 1 < kt_end) issue(smem + (cur ^ 1) * BUF_BYTES, kt + 1);
        GSTAMP();
        const unsigned char* buf = smem + cur * BUF_BYTES;
        tile_mma<T, ATR, BTR, BK, SPLITK, WM, RS>(buf + pa * PANEL_BYTES, oa, buf + (CF::PA + pb) * PANEL_BYTES, ob, acc, accb, do_bsum, lane, &rstat);
        GSTAMP();
        __syncthreads();
        GSTAMP();
        cur ^= 1;
      }
    } else {
    // NST-deep ring, one raw s_barrier per k-tile.  Iteration kt: request k-tile kt + NST - 1 into the buffer everybody left at the
    // last barrier, run the MFMAs of k-tile kt, THEN wait until this wave's pieces of k-tile kt + 1 have landed (counted vmcnt: the
    // NLD * (NST - 2) youngest requests stay in flight) and meet the other waves.  Loads are retired in order, so whatever was
    // requested before the loop (the epilogue's operands) is older than every count used here.
    constexpr int NLD = (CF::PA + CF::PB) * (PANEL * BK * 2 / 1024) / CF::NW;      // LDS-DMA instructions per lane and k-tile
    static_assert(NST == 2 || NST == 3, "ring depth");
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (kt_beg + s < kt_end) issue(smem + s * BUF_BYTES, kt_beg + s);
    if (NST == 3 && kt_beg + 1 < kt_end) wait_vm<NLD>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int nxt = NST - 1;                                 // buffer of k-tile kt + NST - 1
    GSTAMP(); GSTAMP();
    for (int kt = kt_beg; kt < kt_end; ++kt) {
      if (kt + NST - 1 < kt_end) issue(smem + nxt * BUF_BYTES, kt + NST - 1);
      GSTAMP();
      const unsigned buf = lds0 + cur * BUF_BYTES;
      tile_mma_dma<ATR, BTR, BK, SPLITK, WM, RS>(buf + pa * PANEL_BYTES, oa, buf + (CF::PA + pb) * PANEL_BYTES, ob, acc, accb, do_bsum, lane, &rstat);
      GSTAMP();
      if (NST == 3 && kt + 2 < kt_end) wait_vm<NLD>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      GSTAMP();
      cur = (cur + 1 == NST) ? 0 : cur + 1;
      nxt = (nxt + 1 == NST) ? 0 : nxt + 1;
    }
    }
  } else {
    constexpr int NCH = PANEL_BYTES / 16 / 256;
    uint4 ra[NCH], rb[NCH];
    auto fetch = [&](int kt) {
      const T *A, *Bw; int kred, k0;
      src_of(kt, &A, &Bw, &kred, &k0);
      stage_load<T, ATR, BK, NCH>(ra, A, g.lda, M, kred, m0, k0, tid);
      stage_load<T, BTR, BK, NCH>(rb, Bw, g.ldb, N, kred, n0, k0, tid);
    };
    fetch(kt_beg);
    stage_store<T, ATR, BK, NCH>(smem, ra, tid);
    stage_store<T, BTR, BK, NCH>(smem + PANEL_BYTES, rb, tid);
    __syncthreads();
    for (int kt = kt_beg; kt < kt_end; ++kt) {
      const bool has_next = kt + 1 < kt_end;
      if (has_next) fetch(kt + 1);                      // global loads of the next k-tile fly under this tile's MFMAs
      const unsigned char* buf = smem + cur * BUF_BYTES;
      tile_mma<T, ATR, BTR, BK, SPLITK, WM, RS>(buf, oa, buf + PANEL_BYTES, ob, acc, accb, do_bsum, lane, &rstat);
      if (has_next) {
        unsigned char* dA = smem + (cur ^ 1) * BUF_BYTES;
        stage_store<T, ATR, BK, NCH>(dA, ra, tid);
        stage_store<T, BTR, BK, NCH>(dA + PANEL_BYTES, rb, tid);
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  if constexpr (SPLITK) {
    // partial tile -> this split's slab (plain stores; summed by splitk_reduce_kernel)
    float* slab = g.ws + (int64_t)(g.slab_base[pi] + split) * g.slab_stride;
#pragma unroll
    for (int ti = 0; ti < WM; ++ti) {
      const int m = m0 + ra0 + ti * 16 + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int n = n0 + cb0 + tj * 16 + (lane >> 4) * 4;
        if (n >= N) continue;
        *reinterpret_cast<float4*>(slab + (int64_t)m * g.ldc + n) = make_float4(acc[ti][tj][0], acc[ti][tj][1], acc[ti][tj][2], acc[ti][tj][3]);
      }
      if (do_bsum && lane < 16) slab[(int64_t)M * g.ldc + m] = accb[ti][0];
    }
  } else {
    GSTAMP();
    if constexpr (LNP) {
      // LayerNorm-folded forward: the GELU' operand slot (aux) carries colsum(W'), which the epilogue reads itself
      rstat.finish(P.Kred, g.ln_eps);
      epi.prefetch(P, LMV_ACT_NONE, N, g.ldc, m0, n0, wm, wn, lane);
      epi.template store<true>(smem, acc, P, g.act & 0xff, N, g.ldc, m0, n0, wm, wn, lane, wave, rstat.s, rstat.q);
    } else {
      if constexpr (!EPI_EARLY) epi.prefetch(P, g.act & 0xff, N, g.ldc, m0, n0, wm, wn, lane);
      GSTAMP();
#ifdef LMV_GEMM_TIMING
      auto est = [&]() { GSTAMP(); };
      epi.template store<false, decltype(est)>(smem, acc, P, g.act & 0xff, N, g.ldc, m0, n0, wm, wn, lane, wave, nullptr, nullptr, est);
#else
      epi.store(smem, acc, P, g.act & 0xff, N, g.ldc, m0, n0, wm, wn, lane, wave);
#endif
      GSTAMP();
    }
  }
}

// out[i] += sum_s ws[s][i]  (i < nw: dW; nw <= i < nw + nb: db)
// A block covers 256 / SL float4 elements x SL slab lanes: lane l sums slabs l, l + SL, ... in order, lane 0 then adds
// the SL lane sums in order (fixed summation tree: run-to-run reproducible).  SL > 1 keeps small dW matrices with
// hundreds of slabs from being a latency-bound serial walk.
template <int SL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int nslabs, int64_t stride, float* __restrict__ out_w,
                                                           int64_t nw, float* __restrict__ out_b, int nb) {
  constexpr int EL = 256 / SL;
  __shared__ float4 red[SL > 1 ? 256 : 1];
  const int64_t n4 = (nw + (out_b ? nb : 0)) >> 2;
  const int e = threadIdx.x % EL, sl = threadIdx.x / EL;
  const int64_t i = (int64_t)blockIdx.x * EL + e;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float* p = ws + i * 4;
#pragma unroll 8
    for (int s = sl; s < nslabs; s += SL) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)s * stride);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (SL > 1) {
    red[threadIdx.x] = a;
    __syncthreads();
    if (sl != 0) return;
    for (int l = 1; l < SL; ++l) { const float4 v = red[l * EL + e]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  }
  if (i >= n4) return;
  float* o = (i * 4 < nw) ? out_w + i * 4 : out_b + (i * 4 - nw);
  float4 c = *reinterpret_cast<float4*>(o);
  c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w;
  *reinterpret_cast<float4*>(o) = c;
}

// The same reduction for up to LMV_REDUCE_MAX_SEGS weight gradients in ONE launch (lmv_reduce_batch): a block-backward pass used to
// end every dW GEMM with its own ~7 us reduce launch (~160 per train step, mostly launch ramp and tail); the GEMMs of a block now
// leave their slabs in distinct workspace regions and one launch sums them all.  Same fixed summation tree per element as
// splitk_reduce_kernel<SL> (sl slab lanes per element, chosen per segment), so results are bit-identical to the per-GEMM path.
struct ReduceBatch { ReduceSegDev s[LMV_REDUCE_MAX_SEGS]; int n; };

// one block of a slab reduction (the summation tree of splitk_reduce_kernel<SL>, SL per segment); red: 256 float4 of LDS
__device__ __forceinline__ void reduce_slabs_block(const ReduceSegDev& g, int blk, float4* red) {
  const int SL = g.sl, EL = 256 / SL;
  const int64_t n4 = (g.nw + (g.out_b ? g.nb : 0)) >> 2;
  const int e = threadIdx.x % EL, sl = threadIdx.x / EL;
  const int64_t i = (int64_t)blk * EL + e;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float* p = g.ws + i * 4;
#pragma unroll 8
    for (int s = sl; s < g.nslabs; s += SL) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)s * g.stride);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (SL > 1) {                                   // block-uniform
    red[threadIdx.x] = a;
    __syncthreads();
    if (sl != 0) return;
    for (int l = 1; l < SL; ++l) { const float4 v = red[l * EL + e]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  }
  if (i >= n4) return;
  float* o = (i * 4 < g.nw) ? g.out_w + i * 4 : g.out_b + (i * 4 - g.nw);
  float4 c = *reinterpret_cast<float4*>(o);
  c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w;
  *reinterpret_cast<float4*>(o) = c;
}

__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const ReduceBatch rb) {
  __shared__ float4 red[256];
  int si = 0;
#pragma unroll
  for (int k = 1; k < LMV_REDUCE_MAX_SEGS; ++k) if (k < rb.n && (int)blockIdx.x >= rb.s[k].blk0) si = k;
  const ReduceSegDev& g = rb.s[si];
  if (g.kind == LMV_REDUCE_ROWS) {              // block-uniform: partial rows of a column reduction (LayerNorm dgamma | dbeta, dwconv tap sums)
    lmv_partial_reduce_block(red, (int)blockIdx.x - g.blk0, g.ws, g.nslabs, (int)g.stride, g.out_w, (int)g.nw, g.out_b, g.mode);
    return;
  }
  reduce_slabs_block(g, (int)blockIdx.x - g.blk0, red);
}

static int reduce_lanes(int64_t n4, int nslabs) {      // slab lanes: enough blocks to fill the chip, and at most ~32 serial slab reads per thread
  int sl = 1;
  while (sl < 16 && (n4 * sl < 256 * 256 || nslabs > 32 * sl) && nslabs >= 4 * sl) sl *= 4;
  return sl;
}

enum Mode { MODE_FWD = 0, MODE_DX = 1, MODE_DW = 2 };
enum Tile { TILE_128 = 0, TILE_128W8 = 3 };

struct Plan { GemmArgs g; int total, splits, bk, tile, dma, nsplit[2]; size_t ws_bytes; };

int make_plan(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, Mode mode, Plan* pl) {
  if (nproblems < 1 || nproblems > 2) LMV_FAIL(LMV_ERR_SHAPE, "linear: nproblems must be 1 or 2 (got %d)", nproblems);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "linear: unsupported dtype %d", dtype);
  if (N <= 0 || K <= 0 || (N % 8) || (K % 8)) LMV_FAIL(LMV_ERR_SHAPE, "linear: N=%d, K=%d must be positive multiples of 8", N, K);
  GemmArgs& g = pl->g;
  g = GemmArgs{};
  g.nprob = nproblems; g.act = act;
  int out_cols;
  switch (mode) {
    case MODE_FWD: g.N = N; g.lda = K; g.ldb = K; g.ldc = N; out_cols = N; break;
    case MODE_DX:  g.N = K; g.lda = N; g.ldb = K; g.ldc = K; out_cols = K; break;
    default:       g.N = K; g.lda = N; g.ldb = K; g.ldc = K; out_cols = K; break;
  }
  int64_t max_m = 0;
  int min_kred = 1 << 30;
  bool all64 = true, all32 = true;
  for (int i = 0; i < nproblems; ++i) {
    const lmv_linear_problem& q = p[i];
    if (q.rows <= 0 || q.rows > 0x7fffffffLL / 4) LMV_FAIL(LMV_ERR_SHAPE, "linear: bad rows %lld", (long long)q.rows);
    if (!q.a || !q.w || !q.out) LMV_FAIL(LMV_ERR_SHAPE, "linear: null operand");
    if (!lmv_aligned16(q.a) || !lmv_aligned16(q.w) || !lmv_aligned16(q.out) || !lmv_aligned16(q.res) || !lmv_aligned16(q.aux) ||
        !lmv_aligned16(q.out_pre) || !lmv_aligned16(q.bias) || !lmv_aligned16(q.bias_grad))
      LMV_FAIL(LMV_ERR_SHAPE, "linear: operands must be 16-byte aligned");
    if (q.row_scale && q.rows_per_sample <= 0) LMV_FAIL(LMV_ERR_SHAPE, "linear: rows_per_sample must be > 0 with row_scale");
    Problem& P = g.p[i];
    P.a = q.a; P.b = q.w; P.bias = q.bias; P.res = q.res; P.row_scale = q.row_scale; P.aux = q.aux;
    P.out = q.out; P.out_pre = q.out_pre; P.bias_grad = q.bias_grad; P.rps = q.rows_per_sample > 0 ? q.rows_per_sample : 1;
    if (mode == MODE_DW) { P.M = N; P.Kred = (int)q.rows; }
    else { P.M = (int)q.rows; P.Kred = (mode == MODE_FWD) ? K : N; }
    if (P.M > max_m) max_m = P.M;
    if (P.Kred < min_kred) min_kred = P.Kred;
    if (P.Kred % 64) all64 = false;
    if (P.Kred % 32) all32 = false;
  }
  const bool bf = dtype == LMV_BF16;
  const LmvConfig& cf = lmv_config();      // A/B switches: read from the environment ONCE at library load (lmv_config_set changes them at run time)
  // bf16: 64-deep k-tiles unless the reduction is a short non-multiple of 64 (C = 96 layers); fp32: 32-deep
  int bk = (bf && (mode == MODE_DW || all64 || min_kred >= 512)) ? 64 : 32;
  // Occupancy beats k-tile depth whenever the launch has enough tiles to put 4 workgroups on every CU: half of a
  // workgroup's life is launch + first-load latency + epilogue + store drain, which only OTHER resident workgroups
  // hide.  The 32-deep variant needs 32 KB of LDS and <= 128 registers (4 per CU) against 64 KB / 160 (2 per CU).
  int64_t tiles128 = 0;
  for (int i = 0; i < nproblems; ++i) tiles128 += (int64_t)((g.p[i].M + 127) / 128) * ((out_cols + 127) / 128);
  if (bf && all32 && mode != MODE_DW && tiles128 >= cf.gemm_bk32_tiles) bk = 32;
  // dW: 32-deep as well (3 workgroups per CU; tools/dw_sweep.py: best or within 5 % of best on every layer shape)
  if (bf && all32 && mode == MODE_DW && cf.dw_bk == 32) bk = 32;
  if (bf && cf.gemm_bk == 32 && all32) bk = 32;
  if (bf && cf.gemm_bk == 64 && all64) bk = 64;
  const bool no_dma = cf.gemm_no_dma != 0;
  const bool dma = bf && !no_dma && (bk == 64 ? all64 : all32);
  // Tile: 128 x 128 everywhere.  (256 x 128 / 256 x 256 tiles and the 128 x 384 "BigK" tile were measured slower on every layer shape of
  // the model in rounds 1 - 2 -- tile quantisation on 636 / 2544-tile launches, one workgroup per CU -- and left the library in round 3;
  // they live on the branch r02-gemm-experiments.)
  int tile = TILE_128;
  // 64-deep k-tiles at FOUR waves per SIMD: the 128x128 tile on 8 waves of 32x64 (2 workgroups of 64 KB per CU).  Every operand
  // row is then a whole 128-byte line per k-tile -- the 32-deep loop asks L2 for half lines, and L2 (82 % busy on the stage-3 fc2
  // shape, tools/pmc_mem.sh) serves a half line in the same slot as a whole one -- at the price of 1.5x the LDS fragment reads.
  // Measured (tools/bench_kernels.py): forward 3-12 % faster on every K % 64 == 0 shape; dX faster only on the launches that
  // cannot fill 4 workgroups per CU (stage 4), slower elsewhere; dW much slower.
  const bool w8 = bf && !no_dma && all64 && cf.gemm_w8 && cf.gemm_bk == 0 &&
                  (mode == MODE_FWD || (mode == MODE_DX && (tiles128 < 512 || cf.gemm_w8 == 2)));
  if (w8) { bk = 64; tile = TILE_128W8; }
  const int bm = 128, bn = 128;
  g.tiles_n = (out_cols + bn - 1) / bn;
  // dW of two problems that accumulate into the same dW / db (x and c rows through shared weights): one reduction
  g.concat = mode == MODE_DW && nproblems == 2 && p[0].out == p[1].out && p[0].bias_grad == p[1].bias_grad;
  int total = 0;
  for (int i = 0; i < nproblems; ++i) {
    Problem& P = g.p[i];
    P.tiles_m = (P.M + bm - 1) / bm;
    P.tile_begin = (g.concat && i == 1) ? 0 : total;
    if (!(g.concat && i == 1)) total += P.tiles_m * g.tiles_n;
  }
  int max_kt = 1, sum_kt = 0;
  for (int i = 0; i < nproblems; ++i) { const int kt = (g.p[i].Kred + bk - 1) / bk; if (kt > max_kt) max_kt = kt; sum_kt += kt; }
  if (g.concat) max_kt = sum_kt;
  int splits = 1;
  g.kt_per_split = max_kt;
  g.ntiles = total; g.nsplits = 1;
  pl->ws_bytes = 0;
  pl->nsplit[0] = pl->nsplit[1] = 1;
  if (mode == MODE_DW) {
    // Split the token reduction so that ONE generation of workgroups fills the chip: slots = CUs x resident workgroups
    // (3 at 32-deep k-tiles, 2 at 64-deep); each split ends in a plain store of its partial tile.
    const int target = cf.dw_target_blocks;
    int per_xcd = target > 0 ? target / 8 : 32 * (bk == 32 && dma ? 3 : 2);
    if (tile != TILE_128) per_xcd /= 2;      // 8-wave workgroups
    // The workgroups of one split run on one XCD (they share the token rows through its L2), so the split count is a
    // multiple of 8 -- every XCD gets the same number -- or, when one split's tiles already overfill an XCD, 4 / 2 / 1
    // with the tiles of a split divided over the XCDs that share it.
    const int max_splits = (max_kt + 7) / 8;   // at least 8 k-tiles per split
    if (per_xcd >= total && max_splits >= 8) {
      splits = 8 * (per_xcd / total);
      if (splits > max_splits) splits = max_splits / 8 * 8;
    } else {
      splits = 4;
      while (splits > 1 && ((int64_t)total * splits > 8 * (int64_t)per_xcd || splits > max_splits)) splits /= 2;
    }
    g.kt_per_split = (max_kt + splits - 1) / splits;
    g.nsplits = splits;
    splits = (max_kt + g.kt_per_split - 1) / g.kt_per_split;      // non-empty splits (= slabs)
    int slabs = 0;
    for (int i = 0; i < nproblems; ++i) {
      if (g.concat && i == 1) { pl->nsplit[1] = 0; g.slab_base[1] = 0; break; }
      const int kt = g.concat ? max_kt : (g.p[i].Kred + bk - 1) / bk;
      pl->nsplit[i] = (kt + g.kt_per_split - 1) / g.kt_per_split;
      g.slab_base[i] = slabs;
      slabs += pl->nsplit[i];
    }
    g.slab_stride = (int64_t)N * K + N;
    pl->ws_bytes = (size_t)slabs * g.slab_stride * sizeof(float);
  }
  g.cumap = cf.gemm_cumap;
#ifdef LMV_GEMM_TIMING
  { const char* e = getenv("LMV_GEMM_DBG_PTR"); g.dbg = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  pl->total = total; pl->splits = splits; pl->bk = bk; pl->tile = tile; pl->dma = dma;
  return LMV_OK;
}

template <typename T, bool ATR, bool BTR, bool SPLITK, int BK, bool DMA, typename CF, int MINW = 1, int NST = 2, bool LNP = false, bool CONV = false>
int launch_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
  constexpr int lds = NST * (CF::PA + CF::PB) * PANEL * BK * (int)sizeof(T);
  auto kern = gemm_kernel<T, ATR, BTR, SPLITK, BK, DMA, CF, MINW, NST, LNP, CONV>;
  // > 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel AND device.  The call is idempotent, so two threads racing
  // through the first launch both make it; the per-device bit only publishes "done" (re-entrant, no lock).
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      LMV_FAIL(LMV_ERR_LAUNCH, "linear: cannot reserve %d bytes of LDS", lds);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, grid, dim3(CF::NTHR), lds, st, g);
  return LMV_OK;
}

template <bool ATR, bool BTR, bool SPLITK>
int launch_mode(const Plan& pl, dim3 grid, bool bf, hipStream_t st) {
  const GemmArgs& g = pl.g;
  if (g.cv.on) {          // the implicit-GEMM convolution (lmv_conv3x3s2_fwd / _dw): bf16 LDS-DMA kernels only
    if constexpr (ATR != BTR) { LMV_FAIL(LMV_ERR_SHAPE, "conv3x3s2: no data-gradient form"); }
    else {
      if (!bf || !pl.dma) LMV_FAIL(LMV_ERR_SHAPE, "conv3x3s2: bf16 with whole k-tiles only (rows and 9 Cin padded to a multiple of 64)");
      if (pl.bk == 32) return launch_one<bf16_t, ATR, BTR, SPLITK, 32, true, C128, 3, 2, false, true>(g, grid, st);
      if (pl.tile == TILE_128W8) return launch_one<bf16_t, ATR, BTR, SPLITK, 64, true, C128w8, 4, 2, false, true>(g, grid, st);
      return launch_one<bf16_t, ATR, BTR, SPLITK, 64, true, C128, 1, 2, false, true>(g, grid, st);
    }
  }
  if (!bf) return launch_one<float, ATR, BTR, SPLITK, 32, false, C128>(g, grid, st);
  if (!pl.dma) return pl.bk == 64 ? launch_one<bf16_t, ATR, BTR, SPLITK, 64, false, C128>(g, grid, st)
                                  : launch_one<bf16_t, ATR, BTR, SPLITK, 32, false, C128>(g, grid, st);
  // 32-deep k-tiles: 32 KB of LDS and (capped by MINW) <= 128 / 168 registers: 4 (fwd, dX) or 3 (dW) workgroups per CU
  if (pl.bk == 32) {
    const LmvConfig& cf = lmv_config();
    if ((SPLITK ? cf.gemm_nst_dw : cf.gemm_nst) == 3) return launch_one<bf16_t, ATR, BTR, SPLITK, 32, true, C128, 3, 3>(g, grid, st);      // 3-deep ring, 48 KB: 3 workgroups per CU (dW: 140 registers cap it at 3 anyway)
    return launch_one<bf16_t, ATR, BTR, SPLITK, 32, true, C128, SPLITK ? 3 : 4>(g, grid, st);
  }
  if (pl.tile == TILE_128W8) return launch_one<bf16_t, ATR, BTR, SPLITK, 64, true, C128w8, 4>(g, grid, st);
  return launch_one<bf16_t, ATR, BTR, SPLITK, 64, true, C128>(g, grid, st);
}

int launch(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream, Mode mode, void* ws, size_t ws_bytes,
           lmv_reduce_seg* segs = nullptr, int* nsegs = nullptr, const ConvGeo* cv = nullptr) {
  Plan pl;
  if (int rc = make_plan(p, nproblems, N, K, act, dtype, mode, &pl)) return rc;      // (validates the operands)
  if (cv) pl.g.cv = *cv;
  if (!cv && mode == MODE_FWD && dtype == LMV_BF16 && lmv_config().gemm_rs && lmv_rs_eligible(p, nproblems, N, K, act, lmv_config().gemm_rs == 2)) {
    lmv_timing_set_kind(LMV_TK_RS_GEMM);          // (the open bracket of lmv_linear_fwd, if the probe is armed)
    return lmv_rs_linear(p, nproblems, N, K, act, (hipStream_t)stream);              // register-stationary kernel (rsgemm.hip)
  }
  if (!cv && mode == MODE_FWD && dtype == LMV_BF16 && lmv_config().gemm_wn && lmv_wn_eligible(p, nproblems, N, K, act, lmv_config().gemm_wn == 2)) {
    lmv_timing_set_kind(LMV_TK_WN_GEMM);
    return lmv_wn_linear(p, nproblems, N, K, act, (hipStream_t)stream);              // whole-width kernel (wngemm.hip)
  }
  GemmArgs& g = pl.g;
  if (mode == MODE_DW) {
    if (!ws || ws_bytes < pl.ws_bytes || !lmv_aligned16(ws)) LMV_FAIL(LMV_ERR_WORKSPACE, "linear_dw: workspace %zu < %zu bytes", ws_bytes, pl.ws_bytes);
    g.ws = reinterpret_cast<float*>(ws);
  }
  const bool bf = dtype == LMV_BF16;
  dim3 grid(pl.total);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (mode == MODE_DW) grid.x = g.nsplits >= 8 ? pl.total * g.nsplits : 8 * ((pl.total + 8 / g.nsplits - 1) / (8 / g.nsplits));
  double trows = 0.;
  for (int i = 0; i < nproblems && i < 2; ++i) trows += (double)p[i].rows;
  if (mode == MODE_FWD) rc = launch_mode<false, false, false>(pl, grid, bf, st);
  else if (mode == MODE_DX) {          // (timing probe: the transpose-read dX kernel; bytes: dY in, dX out (+ the GELU' operand), W once)
    LmvTimedLaunch timed(stream, 2.0 * N * K * trows, 2.0 * trows * (N + K * (1.0 + (act == LMV_ACT_GELU_GRAD))) + 2.0 * N * K, LMV_TK_GEMM_DX);
    rc = launch_mode<false, true, false>(pl, grid, bf, st);
  } else {                              // (the split-K weight-gradient kernel; bytes: dY and X in, the fp32 slabs out)
    LmvTimedLaunch timed(stream, 2.0 * N * K * trows, 2.0 * trows * (N + K) + 4.0 * N * K * (g.nsplits > 0 ? g.nsplits : 1), LMV_TK_GEMM_DW);
    rc = launch_mode<true, true, true>(pl, grid, bf, st);
  }
  if (rc) return rc;
  LMV_CHECK_LAUNCH("linear");
  if (mode == MODE_DW) {
    LmvTimedLaunch timed_r(stream, 0.0, 4.0 * N * K * ((g.nsplits > 0 ? g.nsplits : 1) + 1.0), LMV_TK_SPLITK_REDUCE);          // (skipped when the reduction is deferred: then the bracket is empty)
    const int64_t nw = (int64_t)N * K;
    const bool shared = nproblems == 2 && p[0].out == p[1].out;
    for (int i = 0; i < nproblems; ++i) {
      if (shared && i == 1) break;
      const int nslabs = shared ? pl.nsplit[0] + pl.nsplit[1] : pl.nsplit[i];
      if (shared && p[0].bias_grad != p[1].bias_grad) LMV_FAIL(LMV_ERR_SHAPE, "linear_dw: problems sharing dW must share db");
      const float* base = g.ws + (int64_t)g.slab_base[i] * g.slab_stride;
      const int64_t n4 = (nw + (p[i].bias_grad ? N : 0)) / 4;
      float* outw = reinterpret_cast<float*>(p[i].out);
      if (segs) {                                  // deferred: the caller sums the slabs later with lmv_reduce_batch
        lmv_reduce_seg& sg = segs[(*nsegs)++];
        sg.ws = base; sg.nslabs = nslabs; sg.slab_stride = g.slab_stride; sg.out_w = outw; sg.nw = nw; sg.out_b = p[i].bias_grad; sg.nb = N;
        sg.kind = LMV_REDUCE_SLABS; sg.mode = 0;
        continue;
      }
      const int sl = reduce_lanes(n4, nslabs);
      const int blocks = (int)((n4 * sl + 255) / 256);
      if (sl == 1) hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks), dim3(256), 0, st, base, nslabs, g.slab_stride, outw, nw, p[i].bias_grad, N);
      else if (sl == 4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, st, base, nslabs, g.slab_stride, outw, nw, p[i].bias_grad, N);
      else hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3(blocks), dim3(256), 0, st, base, nslabs, g.slab_stride, outw, nw, p[i].bias_grad, N);
    }
    LMV_CHECK_LAUNCH("linear_dw reduce");
  }
  return LMV_OK;
}

// LayerNorm-folded forward (lmv_ln_linear_fwd): the forward plan on the 128 x 128 tile, with the row statistics taken from the token
// fragments inside the k-loop and the LayerNorm applied to the accumulators (gemm_kernel<..., LNP = true>)
int launch_ln(const lmv_linear_problem* p, int nproblems, int N, int K, float eps, int act, int dtype, void* stream) {
  Plan pl;
  if (int rc = make_plan(p, nproblems, N, K, act, dtype, MODE_FWD, &pl)) return rc;
  for (int i = 0; i < nproblems; ++i)
    if (!p[i].aux || !p[i].bias) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_fwd: aux (colsum of the folded weight) and bias (folded bias) are required");
  if (!(eps > 0.f)) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_fwd: eps must be > 0");
  GemmArgs& g = pl.g;
  g.ln_eps = eps;
  const dim3 grid(pl.total);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dtype != LMV_BF16) rc = launch_one<float, false, false, false, 32, false, C128, 1, 2, true>(g, grid, st);
  else if (!pl.dma) rc = pl.bk == 64 ? launch_one<bf16_t, false, false, false, 64, false, C128, 1, 2, true>(g, grid, st)
                                     : launch_one<bf16_t, false, false, false, 32, false, C128, 1, 2, true>(g, grid, st);
  else if (pl.bk == 32) rc = launch_one<bf16_t, false, false, false, 32, true, C128, 3, 2, true>(g, grid, st);
  else if (pl.tile == TILE_128W8) rc = launch_one<bf16_t, false, false, false, 64, true, C128w8, 4, 2, true>(g, grid, st);
  else rc = launch_one<bf16_t, false, false, false, 64, true, C128, 1, 2, true>(g, grid, st);
  if (rc) return rc;
  LMV_CHECK_LAUNCH("ln_linear");
  return LMV_OK;
}

}  // namespace

extern "C" int lmv_ln_linear_fwd(const lmv_linear_problem* p, int nproblems, int N, int K, float eps, int act, int dtype, void* stream) {
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_fwd: act must be NONE or GELU");
  return launch_ln(p, nproblems, N, K, eps, act, dtype, stream);
}
// the output projection of an attention module with the block's residual add and DropPath in the GEMM epilogue
extern "C" int lmv_attn_out_proj_residual(const lmv_linear_problem* p, int nproblems, int C, int dtype, void* stream) {
  for (int i = 0; i < nproblems && i < 2; ++i)
    if (!p[i].res) LMV_FAIL(LMV_ERR_SHAPE, "attn_out_proj_residual: the residual operand is required");
  return launch(p, nproblems, C, C, LMV_ACT_NONE, dtype, stream, MODE_FWD, nullptr, 0);
}

extern "C" int lmv_linear_fwd(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream) {
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU && act != LMV_ACT_GELU_GRAD) LMV_FAIL(LMV_ERR_SHAPE, "linear_fwd: act must be NONE, GELU or GELU_GRAD");
  for (int i = 0; i < nproblems && i < 2; ++i)
    if (act == LMV_ACT_GELU_GRAD && (!p[i].aux || p[i].res)) LMV_FAIL(LMV_ERR_SHAPE, "linear_fwd: GELU_GRAD needs aux and takes no residual");
  double rows = 0., bytes = 2.0 * N * K + 4.0 * N;          // algorithmic HBM bytes: A in, C out (+ pre-activation copy out, + residual in, + the GELU' operand in), W and bias once (bf16 = 2 B)
  if (g_lmv_timing_on)
    for (int i = 0; i < nproblems && i < 2; ++i) { rows += (double)p[i].rows; bytes += 2.0 * p[i].rows * (K + (double)N * (1 + (p[i].out_pre != nullptr) + (p[i].res != nullptr) + (p[i].aux != nullptr))); }
  LmvTimedLaunch timed(stream, 2.0 * N * K * rows, bytes);
  return launch(p, nproblems, N, K, act, dtype, stream, MODE_FWD, nullptr, 0);
}
extern "C" int lmv_linear_dx(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream) {
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU_GRAD) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx: act must be NONE or GELU_GRAD");
  for (int i = 0; i < nproblems && i < 2; ++i)
    if (act == LMV_ACT_GELU_GRAD && !p[i].aux) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx: GELU_GRAD needs aux");
  return launch(p, nproblems, N, K, act, dtype, stream, MODE_DX, nullptr, 0);
}
extern "C" size_t lmv_linear_dw_workspace_bytes(const lmv_linear_problem* p, int nproblems, int N, int K, int dtype) {
  Plan pl;
  if (make_plan(p, nproblems, N, K, LMV_ACT_NONE, dtype, MODE_DW, &pl)) return 0;
  return pl.ws_bytes;
}
extern "C" int lmv_linear_dw(const lmv_linear_problem* p, int nproblems, int N, int K, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  return launch(p, nproblems, N, K, LMV_ACT_NONE, dtype, stream, MODE_DW, workspace, workspace_bytes);
}

extern "C" int lmv_linear_dw_partial(const lmv_linear_problem* p, int nproblems, int N, int K, void* workspace, size_t workspace_bytes, int dtype, void* stream,
                                     lmv_reduce_seg* segs, int* nsegs) {
  if (!segs || !nsegs) LMV_FAIL(LMV_ERR_SHAPE, "linear_dw_partial: segs / nsegs must not be NULL");
  *nsegs = 0;
  return launch(p, nproblems, N, K, LMV_ACT_NONE, dtype, stream, MODE_DW, workspace, workspace_bytes, segs, nsegs);
}

// ---- Conv2d(Cin, Cout, 3, stride 2, padding 1) on an NHWC map as an implicit GEMM (models/lemevit.py:701-703, :714-717) ----------------------------------------------
#ifndef LMV_TRY
#define LMV_TRY(expr) do { const int rc__ = (expr); if (rc__) return rc__; } while (0)
#endif
namespace {
int conv_geo(ConvGeo* cv, const void* x, int B, int H, int W, int Cin, int Cout, int KP, int dtype) {
  if (dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "conv3x3s2: bf16 only (fp32: lmv_im2col3x3s2_nhwc + lmv_linear_*)");
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 8) || (Cout % 8) || KP < 9 * Cin || (KP % 64)) LMV_FAIL(LMV_ERR_SHAPE, "conv3x3s2: Cin, Cout multiples of 8, KP >= 9 Cin a multiple of 64");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t rows = (int64_t)B * Ho * Wo;
  if (rows % 64) LMV_FAIL(LMV_ERR_SHAPE, "conv3x3s2: B Ho Wo = %lld must be a multiple of 64 (whole k-tiles of the weight gradient)", (long long)rows);
  if (rows >= (1 << 22) || KP >= (1 << 13) || Ho * Wo >= (1 << 13) * 64 || (int64_t)B * H * W * Cin >= (1LL << 31)) LMV_FAIL(LMV_ERR_SHAPE, "conv3x3s2: map too large for the 32-bit index arithmetic");
  if (!x || !lmv_aligned16(x)) LMV_FAIL(LMV_ERR_SHAPE, "conv3x3s2: null / misaligned map");
  *cv = ConvGeo{};
  cv->on = 1; cv->H = H; cv->W = W; cv->Cin = Cin; cv->Ho = Ho; cv->Wo = Wo; cv->HoWo = Ho * Wo; cv->x = x;
  auto magic = [](unsigned d) { return ((1ull << 40) + d - 1) / d; };
  cv->m_cin = magic((unsigned)Cin); cv->m_wo = magic((unsigned)Wo); cv->m_howo = magic((unsigned)(Ho * Wo));
  return LMV_OK;
}
}  // namespace
extern "C" int lmv_conv3x3s2_fwd(const void* x, const void* wm, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int KP, int act, int dtype, void* stream) {
  ConvGeo cv;
  LMV_TRY(conv_geo(&cv, x, B, H, W, Cin, Cout, KP, dtype));
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU) LMV_FAIL(LMV_ERR_SHAPE, "conv3x3s2_fwd: act must be NONE or GELU");
  lmv_linear_problem p{};
  p.a = x; p.w = wm; p.out = y; p.bias = bias; p.rows = (int64_t)B * cv.HoWo;          // (a: validated only -- the kernel gathers from cv.x)
  const double rows = (double)p.rows;
  LmvTimedLaunch timed(stream, 2.0 * Cout * 9.0 * Cin * rows, 2.0 * ((double)B * H * W * Cin + rows * Cout + (double)Cout * KP));
  return launch(&p, 1, Cout, KP, act, dtype, stream, MODE_FWD, nullptr, 0, nullptr, nullptr, &cv);
}
extern "C" size_t lmv_conv3x3s2_dw_workspace_bytes(int B, int H, int W, int Cin, int Cout, int KP, int dtype) {
  (void)Cin;
  lmv_linear_problem p{};
  p.a = p.w = p.out = (void*)16; p.rows = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  return lmv_linear_dw_workspace_bytes(&p, 1, Cout, KP, dtype);
}
extern "C" int lmv_conv3x3s2_dw(const void* dy, const void* x, float* dwm, float* dbias, int B, int H, int W, int Cin, int Cout, int KP, void* workspace, size_t workspace_bytes, int dtype,
                                void* stream) {
  ConvGeo cv;
  LMV_TRY(conv_geo(&cv, x, B, H, W, Cin, Cout, KP, dtype));
  lmv_linear_problem p{};
  p.a = dy; p.w = x; p.out = dwm; p.bias_grad = dbias; p.rows = (int64_t)B * cv.HoWo;
  return launch(&p, 1, Cout, KP, LMV_ACT_NONE, dtype, stream, MODE_DW, workspace, workspace_bytes, nullptr, nullptr, &cv);
}

extern "C" int lmv_reduce_batch(const lmv_reduce_seg* segs, int nsegs, void* stream) {
  if (nsegs <= 0) return LMV_OK;
  if (!segs) LMV_FAIL(LMV_ERR_SHAPE, "reduce_batch: segs is NULL");
  hipStream_t st = (hipStream_t)stream;
  for (int s0 = 0; s0 < nsegs; s0 += LMV_REDUCE_MAX_SEGS) {
    ReduceBatch rb{};
    rb.n = nsegs - s0 < LMV_REDUCE_MAX_SEGS ? nsegs - s0 : LMV_REDUCE_MAX_SEGS;
    int blocks = 0;
    for (int i = 0; i < rb.n; ++i) {
      const lmv_reduce_seg& q = segs[s0 + i];
      if (!q.ws || !q.out_w || q.nslabs <= 0 || q.nw <= 0 || (q.nw % 4) || (q.out_b && (q.nb % 4)) || !lmv_aligned16(q.ws) || !lmv_aligned16(q.out_w) ||
          !lmv_aligned16(q.out_b) || (q.slab_stride % 4) || (q.kind != LMV_REDUCE_SLABS && q.kind != LMV_REDUCE_ROWS))
        LMV_FAIL(LMV_ERR_SHAPE, "reduce_batch: bad segment %d", s0 + i);
      const int64_t n4 = (q.nw + (q.out_b ? q.nb : 0)) / 4;
      ReduceSegDev& d = rb.s[i];
      d.ws = q.ws; d.out_w = q.out_w; d.out_b = q.out_b; d.stride = q.slab_stride; d.nw = q.nw; d.nslabs = q.nslabs; d.nb = q.nb;
      d.kind = q.kind; d.mode = q.mode;
      d.blk0 = blocks;
      if (q.kind == LMV_REDUCE_ROWS) {           // 32 columns per block (partial_reduce_kernel's geometry); the row width is the slab stride
        if (!q.out_b || q.slab_stride > 0x7fffffff || q.nw > 0x7fffffff || (q.mode != 0 && q.mode != 1) ||
            (q.mode == 0 && q.slab_stride != q.nw + q.nb) || (q.mode == 1 && q.slab_stride != 10 * q.nw))
          LMV_FAIL(LMV_ERR_SHAPE, "reduce_batch: bad partial-row segment %d", s0 + i);
        d.sl = 1;
        blocks += (int)((q.slab_stride / 4 + 7) / 8);
        continue;
      }
      d.sl = reduce_lanes(n4, q.nslabs);
      blocks += (int)((n4 * d.sl + 255) / 256);
    }
    hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3(blocks), dim3(256), 0, st, rb);
  }
  LMV_CHECK_LAUNCH("reduce_batch");
  return LMV_OK;
}
