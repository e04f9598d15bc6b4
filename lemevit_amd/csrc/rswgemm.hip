// rswgemm.hip -- "resident-weight" streaming Linear with an optional LayerNorm prologue for the C = 96 layers (round 3):
//
//   y[r, n] = sum_k LN(x)[r, k] W[n, k] + bias[n]          K = 96, N a multiple of 32 and <= 384, bf16
//
// Which launches: norm1 -> qkv1 / qkv2 of the stage-1 "D" blocks and norm1 -> kv of the stage-0 "C" blocks (models/lemevit.py:560,599:
// 403 456 rows x 96 channels at batch 128).  On the tile kernels this is a LayerNorm launch (read C, write C per token) and a GEMM launch
// (read C, write 3C) that both run at the speed of HBM; the weight is 55 KB.  Here
//   * the WHOLE weight matrix sits in LDS for the life of the workgroup ([N][192 B], chunk swizzle (n >> 2) & 3 inside every 64-byte
//     k-group: conflict-free for the interleaved row mapping below, brute-forced against the ds_read_b128 lane groups of
//     MI355X_MICROARCH.md) -- no operand ring, no barrier after the first one, no LDS-DMA;
//   * every wave streams its own 32-row panels: the rows come straight from global memory as MFMA operand fragments (2 x 3 x 4 VGPRs),
//     the LayerNorm runs on those registers (two-pass fp32 statistics over the
//     24 values a lane holds + two shuffles), the normalised rows go back out (training keeps them for the weight gradient) together with
//     (mean, rstd), and the same registers feed the MFMAs;
//   * output leaves in 32-column slots as 16-byte stores straight from the accumulators (rsw_nmap: two MFMA tiles give a lane 8
//     consecutive columns).
// 16 waves per CU (two 8-wave workgroups of <= 74 KB LDS), plain loads and stores: the compiler counts the waits.
#include <atomic>
#include "common.h"

namespace {

struct RswProb {
  const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* C; bf16_t* ln_out; float* stats;
  int M, wg_begin, nwg, pad_;
};
struct RswArgs {
  RswProb p[2];
  int N, nwg0;
  const float* gamma; const float* beta; float eps;
};

constexpr int RSW_K = 96, RSW_ROWB = RSW_K * 2;      // 192-byte weight rows

// W row (inside a 32-column slot) that MFMA tile t (0 / 1) reads at operand index i (lane & 15): lane group g = lane >> 4 ends up with
// columns 8 g + 4 t + {0..3}: the two tiles of a slot = 8 consecutive columns per lane
__device__ __forceinline__ int rsw_nmap(int t, int i) { return 8 * (i >> 2) + 4 * t + (i & 3); }
__device__ __forceinline__ int rsw_swz(int n) { return (n >> 2) & 3; }

__device__ __forceinline__ f32x4_t rsw_pack8(const float* v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t u = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  return __builtin_bit_cast(f32x4_t, u);
}
__device__ __forceinline__ void rsw_unpack8(const bf16x8_t& c, float* f) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t r = __builtin_bit_cast(u32x4_t, c);
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(r[e] << 16); f[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
}

template <bool LNP>
__global__ __launch_bounds__(512, 4) void rsw_gemm_kernel(const RswArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [N][192 B] weights | bias [N] | gamma [96] | beta [96] (fp32)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool second = (int)blockIdx.x >= g.nwg0;
  RswProb P;
#define RSW_SEL(f) P.f = second ? g.p[1].f : g.p[0].f
  RSW_SEL(A); RSW_SEL(W); RSW_SEL(bias); RSW_SEL(C); RSW_SEL(ln_out); RSW_SEL(stats); RSW_SEL(M); RSW_SEL(wg_begin); RSW_SEL(nwg);
#undef RSW_SEL
  const int N = g.N;
  float* const sbias = reinterpret_cast<float*>(smem + N * RSW_ROWB);
  float* const sgam = sbias + N;
  float* const sbet = sgam + RSW_K;
  // the weight image: logical 16-byte chunk c = 4 ks + q of row n at physical chunk 4 ks + (q ^ swz(n))
  for (int idx = tid; idx < N * 12; idx += 512) {
    const int n = idx / 12, c = idx - n * 12;
    const uint4 v = *reinterpret_cast<const uint4*>(P.W + (long long)n * RSW_K + c * 8);
    *reinterpret_cast<uint4*>(smem + n * RSW_ROWB + (((c & ~3) | ((c & 3) ^ rsw_swz(n))) << 4)) = v;
  }
  for (int i = tid; i < N; i += 512) sbias[i] = P.bias ? P.bias[i] : 0.f;
  if (LNP && tid < RSW_K) { sgam[tid] = g.gamma[tid]; sbet[tid] = g.beta[tid]; }
  __syncthreads();

  const int npanels = (P.M + 31) / 32;
  const int gw = ((int)blockIdx.x - P.wg_begin) * 8 + wave, stride = P.nwg * 8;
  const int grp = lane >> 4, li = lane & 15;
  unsigned woff[2];                                          // byte offsets of the lane's two weight fragments inside a slot (k-step 0)
#pragma unroll
  for (int t = 0; t < 2; ++t) { const int n = rsw_nmap(t, li); woff[t] = n * RSW_ROWB + ((grp ^ rsw_swz(n)) << 4); }

  auto load_panel = [&](int pn, bf16x8_t (&af)[2][3]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = min(pn * 32 + rt * 16 + li, P.M - 1);
      const bf16_t* p = P.A + (long long)row * RSW_K + grp * 8;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) af[rt][ks] = *reinterpret_cast<const bf16x8_t*>(p + ks * 32);
    }
  };
  // (16 waves per CU with 6 KB of rows in flight each hide the HBM latency; a register double buffer for the next panel spilled at 4 waves per SIMD)
  bf16x8_t cur[2][3];
  for (int pn = gw; pn < npanels; pn += stride) {
    load_panel(pn, cur);
    const int row0 = pn * 32 + li;
    if constexpr (LNP) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        float v[3][8];
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          rsw_unpack8(cur[rt][ks], v[ks]);
#pragma unroll
          for (int e = 0; e < 8; ++e) s += v[ks][e];
        }
        s = lmv_xsum4(s);
        const float mean = s * (1.f / RSW_K);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mean; q = fmaf(d, d, q); }
        q = lmv_xsum4(q);
        const float rstd = rsqrtf(q * (1.f / RSW_K) + g.eps);
        const int row = row0 + rt * 16;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int k0 = ks * 32 + grp * 8;
          const float4 g0 = *reinterpret_cast<const float4*>(sgam + k0), g1 = *reinterpret_cast<const float4*>(sgam + k0 + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(sbet + k0), b1 = *reinterpret_cast<const float4*>(sbet + k0 + 4);
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v[ks][e] - mean) * rstd * gg[e] + bb[e];
          const f32x4_t pk = rsw_pack8(o);
          cur[rt][ks] = __builtin_bit_cast(bf16x8_t, pk);
          if (P.ln_out && row < P.M) *reinterpret_cast<f32x4_t*>(P.ln_out + (long long)row * RSW_K + k0) = pk;
        }
        if (P.stats && grp == 0 && row < P.M) { P.stats[2 * (long long)row] = mean; P.stats[2 * (long long)row + 1] = rstd; }
      }
    }
    for (int slot = 0; slot < N / 32; ++slot) {
      f32x4_t acc[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[rt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const unsigned char* wbase = smem + slot * 32 * RSW_ROWB;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        bf16x8_t wf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) wf[t] = *reinterpret_cast<const bf16x8_t*>(wbase + woff[t] + ks * 64);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], cur[rt][ks], acc[rt][t], 0, 0, 0);
      }
      const int c0 = slot * 32 + 8 * grp;
      const float4 b0 = *reinterpret_cast<const float4*>(sbias + c0), b1 = *reinterpret_cast<const float4*>(sbias + c0 + 4);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = row0 + rt * 16;
        const float v[8] = {acc[rt][0][0] + b0.x, acc[rt][0][1] + b0.y, acc[rt][0][2] + b0.z, acc[rt][0][3] + b0.w,
                            acc[rt][1][0] + b1.x, acc[rt][1][1] + b1.y, acc[rt][1][2] + b1.z, acc[rt][1][3] + b1.w};
        if (row < P.M) *reinterpret_cast<f32x4_t*>(P.C + (long long)row * N + c0) = rsw_pack8(v);
      }
    }
  }
}

template <bool LNP>
int rsw_launch(const RswArgs& a, int nwg, hipStream_t st) {
  auto kern = rsw_gemm_kernel<LNP>;
  const int lds = a.N * RSW_ROWB + (a.N + 2 * RSW_K) * (int)sizeof(float);
  static std::atomic<unsigned long long> attr_done{0};      // > 64 KiB of dynamic LDS: opt in once per kernel and device (idempotent)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 384 * RSW_ROWB + (384 + 2 * RSW_K) * 4) != hipSuccess)
      LMV_FAIL(LMV_ERR_LAUNCH, "ln_linear_exact: cannot reserve LDS");
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, st, a);
  return LMV_OK;
}

}  // namespace

extern "C" int lmv_ln_linear_exact_fwd_supported(int N, int K, int dtype) { return dtype == LMV_BF16 && K == RSW_K && N >= 32 && N <= 384 && (N % 32) == 0; }

// y = LN(x) W^T + bias with the EXACT LayerNorm (no folding): p[i].a = x (raw rows), p[i].w = W [N, K], p[i].bias, p[i].out = y;
// seg[i].y = LayerNorm output (or NULL), seg[i].stats = (mean, rstd) (or NULL); gamma == NULL: plain Linear (no LayerNorm, seg ignored)
extern "C" int lmv_ln_linear_exact_fwd(const lmv_linear_problem* p, const lmv_ln_segment* seg, int nproblems, int N, int K, const float* gamma, const float* beta,
                                       float eps, int dtype, void* stream) {
  if (!lmv_ln_linear_exact_fwd_supported(N, K, dtype)) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_exact: N=%d K=%d dtype=%d (bf16, K = 96, N %% 32 == 0, N <= 384)", N, K, dtype);
  if (nproblems < 1 || nproblems > 2 || !p) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_exact: nproblems must be 1 or 2");
  double trows = 0., tbytes = 2.0 * N * K + 4.0 * N;          // raw rows in; LayerNorm(rows), (mean, rstd) and the projection out
  if (g_lmv_timing_on)
    for (int i = 0; i < nproblems; ++i) { trows += (double)p[i].rows; tbytes += 2.0 * p[i].rows * (2.0 * K + N) + 8.0 * p[i].rows; }
  LmvTimedLaunch timed(stream, 2.0 * N * K * trows, tbytes, LMV_TK_RSW_GEMM);
  const bool ln = gamma != nullptr;
  if (ln && (!beta || !(eps > 0.f) || !lmv_aligned16(gamma) || !lmv_aligned16(beta))) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_exact: gamma, beta and eps > 0 are required together");
  RswArgs a{};
  int nwg = 0;
  for (int i = 0; i < nproblems; ++i) {
    const lmv_linear_problem& q = p[i];
    if (q.rows <= 0 || q.rows > 0x7fffffffLL / 2048) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_exact: bad rows %lld", (long long)q.rows);
    if (!q.a || !q.w || !q.out || q.res || q.aux || q.out_pre || q.row_scale) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_exact: a, w, out only (bias optional)");
    if (!lmv_aligned16(q.a) || !lmv_aligned16(q.w) || !lmv_aligned16(q.out) || (seg && !lmv_aligned16(seg[i].y))) LMV_FAIL(LMV_ERR_SHAPE, "ln_linear_exact: operands must be 16-byte aligned");
    RswProb& w = a.p[i];
    w.A = (const bf16_t*)q.a; w.W = (const bf16_t*)q.w; w.bias = q.bias; w.C = (bf16_t*)q.out;
    w.ln_out = (ln && seg) ? (bf16_t*)seg[i].y : nullptr; w.stats = (ln && seg) ? seg[i].stats : nullptr;
    w.M = (int)q.rows;
    // workgroups of 8 waves x 32-row panels: two per CU for the big problem, as many as it has panels for a small one
    const int panels = (int)((q.rows + 31) / 32);
    int want = (panels + 7) / 8;
    if (want > 512) want = 512;
    w.wg_begin = nwg; w.nwg = want;
    nwg += want;
  }
  a.N = N; a.nwg0 = a.p[0].nwg;
  a.gamma = gamma; a.beta = beta; a.eps = eps;
  const int rc = ln ? rsw_launch<true>(a, nwg, (hipStream_t)stream) : rsw_launch<false>(a, nwg, (hipStream_t)stream);
  if (rc) return rc;
  LMV_CHECK_LAUNCH("ln_linear_exact");
  return LMV_OK;
}
