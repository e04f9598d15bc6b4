// NOT PART OF THE LIBRARY (round 3, measured and dropped; kept as a record -- it compiled as lemevit_amd/csrc/wdwgemm.hip against common.h /
// gemm_tiles.h at commit 2d3d5f1 + the dispatch hunk described in DESIGN.md 4.8(i)).  Parity-tested in both orientations; on the stage-3 shapes of
// Base at B = 128 it ran 56.3 / 45.6 / 57.6 us (fc1 / qkv / fc2 weight gradients) against 58.1 / 45.8 / 56.7 us for the 128 x 128 tile kernel, and the
// train step did not move (30.36 vs 30.33 ms, interleaved A/B): 322 MB instead of 484 MB through L2 -> LDS bought nothing, i.e. the weight
// gradient is not bound by that stream at these shapes.
// wdwgemm.hip -- weight-gradient GEMM over whole 384-wide sides (round 3):  dW[n, k] += sum_r dY[r, n] X[r, k]  with N == 384 or K == 384, bf16
//
// The weight gradients of the C = 384 blocks (fc1: 1536 x 384, qkv: 1152 x 384, fc2: 384 x 1536) on 128 x 128 tiles are bound by the
// L2 -> LDS operand stream (64 flop per streamed byte, ~9 TB/s: DESIGN.md 4.8), and on the side stream every microsecond of them costs the
// train step about half a microsecond.  Same answer as csrc/wngemm.hip: one 8-wave workgroup owns a 128-wide strip of the LONG side
// ("P": 128 columns of dY for fc1 / qkv, of X for fc2) and ALL 384 columns of the short side ("Q"): 96 flop per streamed byte, wave tile
// 64 x 96, and the requests are split by wave -- the P strip is read by this workgroup only (HBM: waves 6..7, three 64-row steps ahead
// through four 16 KB slots), the Q panels by every workgroup of the same row range (L2: waves 0..5, one step ahead through two 48 KB
// stages).  Both operands are reduction-major ([rows][columns]): LDS images and ds_read_b64_tr_b16 fragment reads are the transposed-panel
// forms of gemm_tiles.h.  The token reduction is split over workgroups as in gemm.hip (fp32 slabs [N*K | N], summed by
// splitk_reduce_kernel in a fixed order); the bias gradient rides the matrix pipe (ones x dY fragments).
#include <atomic>
#include "common.h"
#include "gemm_tiles.h"

namespace {

struct WdwArgs {
  const bf16_t* P[2]; const bf16_t* Q[2];      // the two row sources (image tokens, meta tokens) of one concatenated reduction
  int kt0, kt_total;                           // 64-row steps of source 0 / of both
  int ldp, pdim, ptiles;                       // P: row stride (= its column count), columns, 128-wide strips
  int nsplits, kt_per_split, total;            // splits of the reduction, steps per split, workgroups = ptiles * nsplits
  int N, K, has_bias;                          // dW is [N, K]
  float* ws; long long slab_stride;            // slabs [nsplits][N * K + N]
};

constexpr int WDW_Q = 384, WDW_BK = 64;
constexpr int WDW_PANEL = PANEL * WDW_BK * 2;                     // one [64 rows x 128 columns] transposed panel: 16 KB
constexpr int WDW_QSTAGE = 3 * WDW_PANEL;                          // 48 KB
constexpr int WDW_P_OFF = 2 * WDW_QSTAGE, WDW_LDS = 2 * WDW_QSTAGE + 4 * WDW_PANEL;      // 160 KB

template <bool SWAP>      // SWAP = false: P = dY (its columns are dW rows), Q = X;  true: P = X (its columns are dW columns), Q = dY
__global__ __launch_bounds__(512, 2) void wdw_gemm_kernel(const WdwArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                         // 2 (P side) x 4 (Q side) waves of 64 x 96
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  // XCD-contiguous order: hardware places block b on XCD b % 8; XCD x works on a contiguous run of (split, strip) pairs, so the
  // workgroups of one split -- they read the same Q rows -- mostly share one L2 (speed only)
  int id;
  {
    const int T_ = g.total, xcd = blockIdx.x & 7, q = T_ >> 3, r = T_ & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int split = id / g.ptiles, pt = id - split * g.ptiles;
  const int kt_beg = split * g.kt_per_split, kt_end = min(g.kt_total, kt_beg + g.kt_per_split);
  if (kt_beg >= kt_end) return;
  const int KS = kt_end - kt_beg;

  const bool a_wave = wave >= 6;                                   // wave-uniform: P pieces (waves 6, 7) or Q pieces (0..5: panel wave >> 1)
  // This wave's eight 1 KiB pieces of a k-step (transposed panel image [64 rows][128 columns], gemm_tiles.h::panel_dma<true>): piece i
  // covers rows 4 sg .. 4 sg + 3, sg = w2 + 2 i (w2 = the wave's index inside its pair); lane -> row r = 4 sg + (lane >> 4), 16-byte
  // chunk c16 = (((p16 >> 1) ^ swz_t(r)) << 1) | (p16 & 1), p16 = lane & 15.  The element offsets are computed once; a step adds 64 rows.
  long long eoff[8];
  const int w2 = a_wave ? wave - 6 : (wave & 1);
  const int ld = a_wave ? g.ldp : WDW_Q, dim = a_wave ? g.pdim : WDW_Q, tile0 = a_wave ? pt * PANEL : (wave >> 1) * PANEL;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int sg = w2 + 2 * i, r = sg * 4 + (lane >> 4), p16 = lane & 15, c16 = ((((p16 >> 1) ^ swz_t(r))) << 1) | (p16 & 1);
    eoff[i] = (long long)r * ld + min(tile0 + c16 * 8, dim - 8);
  }
  const bf16_t* const s0 = a_wave ? g.P[0] : g.Q[0]; const bf16_t* const s1 = a_wave ? g.P[1] : g.Q[1];
  auto issue_step = [&](int kt, int qbuf, int pslot) {
    const bool second = kt >= g.kt0;
    const bf16_t* base = (second ? s1 : s0) + (long long)((second ? kt - g.kt0 : kt) * WDW_BK) * ld;
    unsigned char* panel = a_wave ? smem + WDW_P_OFF + pslot * WDW_PANEL : smem + qbuf * WDW_QSTAGE + (wave >> 1) * WDW_PANEL;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((glb_vp)(base + eoff[i]), (lds_vp)(panel + (w2 + 2 * i) * 1024), 16, 0, 0);
  };

  f32x4_t acc[4][6], accb[SWAP ? 6 : 4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < (SWAP ? 6 : 4); ++i) accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_bsum = g.has_bias && (SWAP ? (pt == 0 && wm == 0) : wn == 0);      // wave-uniform: one wave per dY column covers its bias sum

  // fragment read offsets (k-half 0; half 1: + 32 rows of 256 B = 8192 bytes)
  unsigned offp[4], offq[6];
#pragma unroll
  for (int t = 0; t < 4; ++t) offp[t] = frag_off<true, WDW_BK>(64 * wm + 16 * t, lane, 0);
#pragma unroll
  for (int t = 0; t < 6; ++t) { const int col = 96 * wn + 16 * t; offq[t] = (col >> 7) * WDW_PANEL + frag_off<true, WDW_BK>(col & 127, lane, 0); }

  // prologue: Q of the first step; P of the first three
  if (a_wave) {
    issue_step(kt_beg, 0, 0);
    if (KS > 1) issue_step(kt_beg + 1, 0, 1);
    if (KS > 2) issue_step(kt_beg + 2, 0, 2);
    if (KS > 2) wait_vm<16>(); else if (KS > 1) wait_vm<8>(); else wait_vm<0>();
  } else {
    issue_step(kt_beg, 0, 0);
    wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
  const u16x8_t o16 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, o16);
  int buf = 0;
  for (int ks = 0; ks < KS; ++ks) {
    const unsigned qbase = lds0 + buf * WDW_QSTAGE, pbase = lds0 + WDW_P_OFF + (ks & 3) * WDW_PANEL;
    // this wave's requests up front (a stage / slot nobody reads any more: everybody passed the last barrier)
    if (a_wave) { if (ks + 3 < KS) issue_step(kt_beg + ks + 3, 0, (ks + 3) & 3); }
    else { if (ks + 1 < KS) issue_step(kt_beg + ks + 1, buf ^ 1, 0); }
    bf16x8_t pf[2][4], qf[2][6];
    auto rd = [&](int h) {
#pragma unroll
      for (int t = 0; t < 6; ++t) qf[h][t] = frag_ld<true>(qbase + offq[t] + h * 8192);
#pragma unroll
      for (int t = 0; t < 4; ++t) pf[h][t] = frag_ld<true>(pbase + offp[t] + h * 8192);
    };
    rd(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      wait_lgkm<0>();
#pragma unroll
      for (int t = 0; t < 6; ++t) asm volatile("" : "+v"(qf[h][t]));       // the MFMAs below must not be scheduled above the wait
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(pf[h][t]));
      if (h == 0) rd(1);                                                   // the second half's fragments fly under the first half's MFMAs
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 6; ++tj)
          acc[ti][tj] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[h][ti], qf[h][tj], acc[ti][tj], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[h][tj], pf[h][ti], acc[ti][tj], 0, 0, 0);
      if (do_bsum) {                                                        // column sums of dY = ones^T dY
        if (SWAP) {
#pragma unroll
          for (int tj = 0; tj < 6; ++tj) accb[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, qf[h][tj], accb[tj], 0, 0, 0);
        } else {
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) accb[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[h][ti], accb[ti], 0, 0, 0);
        }
      }
    }
    if (a_wave) {
      if (ks + 3 < KS) wait_vm<16>(); else if (ks + 2 < KS) wait_vm<8>(); else wait_vm<0>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    buf ^= 1;
  }

  // ---- this split's slab: plain float4 stores (the lane owns 4 consecutive dW columns of one dW row) --------------------------------
  float* slab = g.ws + (long long)split * g.slab_stride;
  const int NK = g.N * g.K;
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
    for (int tj = 0; tj < 6; ++tj) {
      if (!SWAP) {      // dW row = P column (n), dW columns = Q columns (k)
        const int n = pt * PANEL + 64 * wm + 16 * ti + (lane & 15), k = 96 * wn + 16 * tj + 4 * (lane >> 4);
        if (n < g.N) *reinterpret_cast<float4*>(slab + (long long)n * WDW_Q + k) = make_float4(acc[ti][tj][0], acc[ti][tj][1], acc[ti][tj][2], acc[ti][tj][3]);
      } else {          // dW row = Q column (n), dW columns = P columns (k)
        const int n = 96 * wn + 16 * tj + (lane & 15), k = pt * PANEL + 64 * wm + 16 * ti + 4 * (lane >> 4);
        if (k < g.K) *reinterpret_cast<float4*>(slab + (long long)n * g.K + k) = make_float4(acc[ti][tj][0], acc[ti][tj][1], acc[ti][tj][2], acc[ti][tj][3]);
      }
    }
  }
  if (do_bsum && lane < 16) {
    if (SWAP) {
#pragma unroll
      for (int tj = 0; tj < 6; ++tj) slab[NK + 96 * wn + 16 * tj + lane] = accb[tj][0];
    } else {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) { const int n = pt * PANEL + 64 * wm + 16 * ti + lane; if (n < g.N) slab[NK + n] = accb[ti][0]; }
    }
  }
}

template <bool SWAP>
int wdw_launch(const WdwArgs& a, hipStream_t st) {
  auto kern = wdw_gemm_kernel<SWAP>;
  static std::atomic<unsigned long long> attr_done{0};      // > 64 KiB of dynamic LDS: opt in once per kernel and device (idempotent)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WDW_LDS) != hipSuccess)
      LMV_FAIL(LMV_ERR_LAUNCH, "wdw linear: cannot reserve %d bytes of LDS", WDW_LDS);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(a.total), dim3(512), WDW_LDS, st, a);
  return LMV_OK;
}

}  // namespace

// Does the whole-side kernel take this weight-gradient launch?  On success *nsplits = slabs it writes ([N*K | N] floats each).
// One dW (a single problem, or two problems accumulating into the same dW / db), one side 384 wide and the other a multiple of 128
// and >= 512, every row count a multiple of 64, a reduction long enough for >= 4 steps per split.
bool lmv_wdw_plan(const lmv_linear_problem* p, int nproblems, int N, int K, bool force, int* nsplits) {
  if (nproblems < 1 || nproblems > 2) return false;
  if (nproblems == 2 && (p[0].out != p[1].out || p[0].bias_grad != p[1].bias_grad)) return false;
  const bool swap = N == WDW_Q;                   // fc2 form: dW rows are the 384-wide side
  const int pdim = swap ? K : N;
  if (!((N == WDW_Q) != (K == WDW_Q)) || pdim < 512 || (pdim % PANEL)) return false;
  long long steps = 0;
  for (int i = 0; i < nproblems; ++i) {
    if (p[i].rows <= 0 || (p[i].rows % WDW_BK) || p[i].rows > 0x7fffffffLL / 2048) return false;
    steps += p[i].rows / WDW_BK;
  }
  const int ptiles = pdim / PANEL;
  int ns = 256 / ptiles;                           // one workgroup (160 KB of LDS) per CU
  if (ns > steps / 4) ns = (int)(steps / 4);
  if (ns < 1) return false;
  if (!force && steps < 128) return false;         // short reductions: the 128 x 128 tiles fill the chip better
  const int per = (int)((steps + ns - 1) / ns);
  *nsplits = (int)((steps + per - 1) / per);
  return true;
}

int lmv_wdw_linear(const lmv_linear_problem* p, int nproblems, int N, int K, float* ws, long long slab_stride, hipStream_t st) {
  int ns = 0;
  if (!lmv_wdw_plan(p, nproblems, N, K, true, &ns)) LMV_FAIL(LMV_ERR_SHAPE, "wdw linear: unsupported launch");
  const bool swap = N == WDW_Q;
  WdwArgs a{};
  long long steps = 0;
  for (int i = 0; i < 2; ++i) {
    const lmv_linear_problem& q = p[i < nproblems ? i : 0];
    a.P[i] = (const bf16_t*)(swap ? q.w : q.a); a.Q[i] = (const bf16_t*)(swap ? q.a : q.w);      // a = dY [rows, N], w = X [rows, K]
    if (i < nproblems) steps += q.rows / WDW_BK;
  }
  a.kt0 = (int)(p[0].rows / WDW_BK); a.kt_total = (int)steps;
  a.pdim = swap ? K : N; a.ldp = a.pdim; a.ptiles = a.pdim / PANEL;
  a.kt_per_split = (int)((steps + ns - 1) / ns); a.nsplits = ns; a.total = a.ptiles * ns;
  a.N = N; a.K = K; a.has_bias = p[0].bias_grad != nullptr;
  a.ws = ws; a.slab_stride = slab_stride;
  const int rc = swap ? wdw_launch<true>(a, st) : wdw_launch<false>(a, st);
  if (rc) return rc;
  LMV_CHECK_LAUNCH("wdw linear");
  return LMV_OK;
}
