// norm.hip -- LayerNorm forward / backward over token rows (memory-bound; graded on HBM GB/s).
//
// A row of C elements is C/8 (bf16) or C/4 (fp32) 16-byte chunks.  LPR = 2^k lanes cooperate on a
// row (k chosen on the host so that chunks/LPR <= NIT with the fewest idle lanes), so a 64-lane
// wavefront processes 64/LPR rows at once and every lane issues 16-byte loads; the row lives in
// registers between the statistics pass and the normalise pass (one HBM read, one HBM write).
// Two-pass (mean, then centred variance) in fp32 to match torch's numerics.
//
// One launch covers up to TWO row segments that share (gamma, beta): the image-token matrix and the
// meta-token matrix of a block are normalised by the same LayerNorm (models/lemevit.py:560-564).
// Backward: dgamma / dbeta are reduced registers -> LDS transpose (fixed-order sums, no atomics or shuffles: the
// shuffle + LDS-atomic version cost 10-20 us per launch) -> one partial row per workgroup; a second tiny kernel
// sums the partial rows (no global atomics: they serialise across XCDs).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int BWD_MAXC = 2048;

template <typename T>
struct Segs {
  const T* x[2]; T* y[2]; float* stats[2]; const T* dy[2]; const T* dres[2]; T* dx[2];
  const float* sc[2]; T* dxs[2]; int64_t rps[2];      // bwd: optional DropPath-scaled second copy of dx
  int64_t rows0, total;
};

__device__ __forceinline__ float group_sum(float v, int lpr) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    if (o < lpr) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int NIT>
__global__ __launch_bounds__(TPB) void ln_fwd_kernel(const Segs<T> sg, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    int C, int lpr_log2, float eps, int act) {
  constexpr int EPC = DT<T>::EPC;
  const int lpr = 1 << lpr_log2, nch = C / EPC;
  const int lir = threadIdx.x & (lpr - 1), rib = threadIdx.x >> lpr_log2, rpb = TPB >> lpr_log2;
  const float invC = 1.f / (float)C;
  // Chunk indices past the row are CLAMPED so every load is unconditional (a bounds branch around each load costs one
  // L2 round trip per load); `vm` zeroes what the clamped lanes would add to the sums, and only the stores are guarded.
  float gm[NIT][EPC], bt[NIT][EPC], vm[NIT];
  int chs[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lir + it * lpr;
    chs[it] = min(ch, nch - 1);
    vm[it] = (ch < nch) ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { gm[it][e] = gamma[chs[it] * EPC + e]; bt[it][e] = beta[chs[it] * EPC + e]; }
  }
  for (int64_t row = (int64_t)blockIdx.x * rpb + rib; row < sg.total; row += (int64_t)gridDim.x * rpb) {
    const int s = row >= sg.rows0;
    const int64_t lr = row - (s ? sg.rows0 : 0);
    const T* xr = (s ? sg.x[1] : sg.x[0]) + lr * C;
    uint4 raw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) raw[it] = *reinterpret_cast<const uint4*>(xr + chs[it] * EPC);
    float v[NIT][EPC];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      chunk_to_f<T>(raw[it], v[it]);
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < EPC; ++e) t += v[it][e];
      sum += t * vm[it];
    }
    const float mean = group_sum(sum, lpr) * invC;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < EPC; ++e) { const float d = v[it][e] - mean; t += d * d; }
      q += t * vm[it];
    }
    const float rstd = 1.f / sqrtf(group_sum(q, lpr) * invC + eps);
    T* yr = (s ? sg.y[1] : sg.y[0]) + lr * C;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      float o[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) o[e] = (v[it][e] - mean) * rstd * gm[it][e] + bt[it][e];
      if (act) {                                           // y = GELU(LN(x)): the meta-token MLPs (models/lemevit.py:731-743)
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = sizeof(T) == 4 ? gelu_f(o[e]) : gelu_fast_f(o[e]);
      }
      if (lir + it * lpr < nch) *reinterpret_cast<uint4*>(yr + chs[it] * EPC) = f_to_chunk<T>(o);
    }
    float* st = s ? sg.stats[1] : sg.stats[0];
    if (st && lir == 0) { st[lr * 2] = mean; st[lr * 2 + 1] = rstd; }
  }
}

// ACT: the forward was y = GELU(LN(x)); dy is first multiplied by GELU'(LN(x)), recomputed from x, gamma and beta
template <typename T, int NIT, bool ACT>
__global__ __launch_bounds__(TPB) void ln_bwd_kernel(const Segs<T> sg, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ partial, int C, int lpr_log2) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ __attribute__((aligned(16))) float red[TPB * 2 * EPC];
  const int lpr = 1 << lpr_log2, nch = C / EPC;
  const int lir = threadIdx.x & (lpr - 1), rib = threadIdx.x >> lpr_log2, rpb = TPB >> lpr_log2;
  const float invC = 1.f / (float)C;
  float adg[NIT][EPC], adb[NIT][EPC], gm[NIT][EPC], bt[ACT ? NIT : 1][EPC], vm[NIT];
  int chs[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lir + it * lpr;
    chs[it] = min(ch, nch - 1);                    // clamped: all loads are unconditional (see ln_fwd_kernel)
    vm[it] = (ch < nch) ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      adg[it][e] = 0.f; adb[it][e] = 0.f; gm[it][e] = gamma[chs[it] * EPC + e];
      if (ACT) bt[it][e] = beta[chs[it] * EPC + e];
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * rpb + rib; row < sg.total; row += (int64_t)gridDim.x * rpb) {
    const int s = row >= sg.rows0;
    const int64_t lr = row - (s ? sg.rows0 : 0);
    const float* st = s ? sg.stats[1] : sg.stats[0];
    const T* xr = (s ? sg.x[1] : sg.x[0]) + lr * C;
    const T* dyr = (s ? sg.dy[1] : sg.dy[0]) + lr * C;
    const T* drp = s ? sg.dres[1] : sg.dres[0];
    T* dxr = (s ? sg.dx[1] : sg.dx[0]) + lr * C;
    const float* dsc = s ? sg.sc[1] : sg.sc[0];
    T* dsp = s ? sg.dxs[1] : sg.dxs[0];
    const float dsc_v = dsc ? dsc[lr / (s ? sg.rps[1] : sg.rps[0])] : 1.f;
    const float mean = st[lr * 2], rstd = st[lr * 2 + 1];
    uint4 rx[NIT], rdy[NIT], rres[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      rx[it] = *reinterpret_cast<const uint4*>(xr + chs[it] * EPC);
      rdy[it] = *reinterpret_cast<const uint4*>(dyr + chs[it] * EPC);
    }
    if (drp) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) rres[it] = *reinterpret_cast<const uint4*>(drp + lr * C + chs[it] * EPC);
    }
    float xh[NIT][EPC], g[NIT][EPC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      float xv[EPC], dv[EPC];
      chunk_to_f<T>(rx[it], xv);
      chunk_to_f<T>(rdy[it], dv);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        dv[e] *= vm[it];
        xh[it][e] = (xv[e] - mean) * rstd;
        if (ACT) {
          const float pre = xh[it][e] * gm[it][e] + bt[it][e];
          dv[e] *= sizeof(T) == 4 ? gelu_grad_f(pre) : gelu_grad_fast_f(pre);
        }
        g[it][e] = dv[e] * gm[it][e];
        s1 += g[it][e]; s2 += g[it][e] * xh[it][e];
        adg[it][e] += dv[e] * xh[it][e]; adb[it][e] += dv[e];
      }
    }
    s1 = group_sum(s1, lpr) * invC; s2 = group_sum(s2, lpr) * invC;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      float o[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) o[e] = rstd * (g[it][e] - s1 - xh[it][e] * s2);
      if (drp) {
        float r[EPC];
        chunk_to_f<T>(rres[it], r);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] += r[e];
      }
      if (lir + it * lpr < nch) {
        *reinterpret_cast<uint4*>(dxr + chs[it] * EPC) = f_to_chunk<T>(o);
        if (dsc) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) o[e] *= dsc_v;
          *reinterpret_cast<uint4*>(dsp + lr * C + chs[it] * EPC) = f_to_chunk<T>(o);
        }
      }
    }
  }
  // Column sums over the row groups of this workgroup, one chunk slot (`it`) at a time: every thread drops its
  // 2 x EPC sums into LDS as [row group][lane in row][dg | db]; output j of the slot is then the sum of `rpb` floats
  // a fixed stride apart (consecutive threads -> consecutive banks).  No atomics, no shuffles, fixed order.
  float* prow = partial + (int64_t)blockIdx.x * 2 * C;
  const int nout = lpr * 2 * EPC;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    __syncthreads();
    float* w = red + threadIdx.x * (2 * EPC);
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      *reinterpret_cast<float4*>(w + e) = make_float4(adg[it][e], adg[it][e + 1], adg[it][e + 2], adg[it][e + 3]);
      *reinterpret_cast<float4*>(w + EPC + e) = make_float4(adb[it][e], adb[it][e + 1], adb[it][e + 2], adb[it][e + 3]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nout; j += TPB) {
      float s = 0.f;
      for (int r = 0; r < rpb; ++r) s += red[r * nout + j];
      const int l = j / (2 * EPC), k = j - l * (2 * EPC), ch = l + it * lpr;
      if (ch < nch) prow[(k < EPC ? 0 : C - EPC) + ch * EPC + k] = s;
    }
  }
}

// lanes per row (log2) and iterations for a row of `nch` 16-byte chunks
inline bool pick_geometry(int nch, int max_it, int* lpr_log2, int* nit) {
  int best = -1, best_score = 1 << 30, best_it = 0;
  for (int l = 2; l <= 6; ++l) {
    const int lpr = 1 << l, it = (nch + lpr - 1) / lpr;
    if (it > max_it) continue;
    const int waste = it * lpr - nch;                            // idle lane-iterations per row
    const int score = waste * 100 + (it > 3 ? it - 3 : 3 - it);  // then ~3 loads in flight per lane
    if (score <= best_score) { best = l; best_score = score; best_it = it; }      // ties: more lanes per row (fewer registers per lane)
  }
  if (best < 0) return false;
  *lpr_log2 = best; *nit = best_it;
  return true;
}

inline int bwd_blocks(int64_t rows, int l2) {
  const int rpb = TPB >> l2;
  const int cap = lmv_config().ln_bwd_blocks, minrows = lmv_config().ln_bwd_minrows;
  int64_t blocks = (rows + minrows * rpb - 1) / (minrows * rpb);           // >= 2 rows per row group
  return (int)(blocks > cap ? cap : (blocks < 1 ? 1 : blocks));
}

template <typename T>
int fill(Segs<T>* sg, const lmv_ln_segment* seg, int nseg, bool bwd, const char* who) {
  if (nseg < 1 || nseg > 2) LMV_FAIL(LMV_ERR_SHAPE, "%s: nseg must be 1 or 2", who);
  *sg = Segs<T>{};
  for (int i = 0; i < nseg; ++i) {
    const lmv_ln_segment& s = seg[i];
    if (s.rows <= 0) LMV_FAIL(LMV_ERR_SHAPE, "%s: segment %d has no rows", who, i);
    if (!s.x || !lmv_aligned16(s.x)) LMV_FAIL(LMV_ERR_SHAPE, "%s: null or misaligned x", who);
    if (!bwd && (!s.y || !lmv_aligned16(s.y))) LMV_FAIL(LMV_ERR_SHAPE, "%s: null or misaligned y", who);
    if (bwd && (!s.dy || !s.dx || !s.stats || !lmv_aligned16(s.dy) || !lmv_aligned16(s.dx) || !lmv_aligned16(s.dres)))
      LMV_FAIL(LMV_ERR_SHAPE, "%s: null or misaligned gradient operand", who);
    sg->x[i] = (const T*)s.x; sg->y[i] = (T*)s.y; sg->stats[i] = s.stats; sg->dy[i] = (const T*)s.dy; sg->dres[i] = (const T*)s.dres; sg->dx[i] = (T*)s.dx;
    if (bwd && (s.dx_scale || s.dx_scaled)) {
      if (!s.dx_scale || !s.dx_scaled || s.rows_per_sample <= 0 || !lmv_aligned16(s.dx_scaled))
        LMV_FAIL(LMV_ERR_SHAPE, "%s: dx_scale, dx_scaled and rows_per_sample come together", who);
      sg->sc[i] = s.dx_scale; sg->dxs[i] = (T*)s.dx_scaled; sg->rps[i] = s.rows_per_sample;
    }
  }
  sg->rows0 = seg[0].rows;
  sg->total = seg[0].rows + (nseg == 2 ? seg[1].rows : 0);
  return LMV_OK;
}

template <typename T>
int launch_fwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, int C, float eps, int act, hipStream_t st) {
  Segs<T> sg;
  if (int rc = fill(&sg, seg, nseg, false, "layernorm")) return rc;
  int l2, nit;
  if (!pick_geometry(C / DT<T>::EPC, sizeof(T) == 4 ? 8 : 6, &l2, &nit)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm: C=%d too wide", C);
  const int rpb = TPB >> l2;
  int64_t blocks = (sg.total + rpb - 1) / rpb; if (blocks > 4096) blocks = 4096;
  dim3 grid((int)blocks), block(TPB);
#define LN_FWD_CASE(N) case N: hipLaunchKernelGGL((ln_fwd_kernel<T, N>), grid, block, 0, st, sg, gamma, beta, C, l2, eps, act); break;
  switch (nit) {
    LN_FWD_CASE(1) LN_FWD_CASE(2) LN_FWD_CASE(3) LN_FWD_CASE(4) LN_FWD_CASE(5) LN_FWD_CASE(6)
    default:
      if constexpr (sizeof(T) == 4) { switch (nit) { LN_FWD_CASE(7) LN_FWD_CASE(8) } }
      else LMV_FAIL(LMV_ERR_SHAPE, "layernorm: C=%d too wide", C);
  }
#undef LN_FWD_CASE
  LMV_CHECK_LAUNCH("layernorm_fwd");
  return LMV_OK;
}

template <typename T, bool ACT>
int launch_bwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, float* dgamma, float* dbeta, int C, void* ws, size_t ws_bytes,
               hipStream_t st, int* partial_rows = nullptr) {
  Segs<T> sg;
  if (int rc = fill(&sg, seg, nseg, true, "layernorm_bwd")) return rc;
  int l2, nit;
  if (C > BWD_MAXC || !pick_geometry(C / DT<T>::EPC, sizeof(T) == 4 ? 8 : 6, &l2, &nit)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: C=%d too wide", C);
  const int blocks = bwd_blocks(sg.total, l2);
  const size_t need = (size_t)blocks * 2 * C * sizeof(float);
  if (!ws || ws_bytes < need) LMV_FAIL(LMV_ERR_WORKSPACE, "layernorm_bwd: workspace %zu < %zu bytes", ws_bytes, need);
  float* partial = reinterpret_cast<float*>(ws);
  dim3 grid(blocks), block(TPB);
#define LN_BWD_CASE(N) case N: hipLaunchKernelGGL((ln_bwd_kernel<T, N, ACT>), grid, block, 0, st, sg, gamma, beta, partial, C, l2); break;
  switch (nit) {
    LN_BWD_CASE(1) LN_BWD_CASE(2) LN_BWD_CASE(3) LN_BWD_CASE(4) LN_BWD_CASE(5) LN_BWD_CASE(6)
    default:
      if constexpr (sizeof(T) == 4) { switch (nit) { LN_BWD_CASE(7) LN_BWD_CASE(8) } }
      else LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: C=%d too wide", C);
  }
#undef LN_BWD_CASE
  LMV_CHECK_LAUNCH("layernorm_bwd");
  if (partial_rows) { *partial_rows = blocks; return LMV_OK; }      // the caller sums the per-workgroup rows (lmv_layernorm_bwd_reduce), e.g. on another stream
  return lmv_launch_partial_reduce(partial, blocks, 2 * C, dgamma, C, dbeta, 0, st);
}

}  // namespace

extern "C" int lmv_layernorm_fwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, int C, float eps, int dtype, void* stream) {
  if (!seg) LMV_FAIL(LMV_ERR_SHAPE, "layernorm: null segments");
  if (C <= 0 || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm: C=%d must be a positive multiple of 8", C);
  if (!gamma || !beta || !lmv_aligned16(gamma) || !lmv_aligned16(beta)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm: null or misaligned affine");
  if (dtype == LMV_BF16) return launch_fwd<bf16_t>(seg, nseg, gamma, beta, C, eps, 0, (hipStream_t)stream);
  if (dtype == LMV_F32) return launch_fwd<float>(seg, nseg, gamma, beta, C, eps, 0, (hipStream_t)stream);
  LMV_FAIL(LMV_ERR_DTYPE, "layernorm: unsupported dtype %d", dtype);
}

extern "C" int lmv_layernorm_gelu_fwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, int C, float eps, int dtype,
                                      void* stream) {
  if (!seg) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_gelu: null segments");
  if (C <= 0 || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_gelu: C=%d must be a positive multiple of 8", C);
  if (!gamma || !beta || !lmv_aligned16(gamma) || !lmv_aligned16(beta)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_gelu: null or misaligned affine");
  if (dtype == LMV_BF16) return launch_fwd<bf16_t>(seg, nseg, gamma, beta, C, eps, 1, (hipStream_t)stream);
  if (dtype == LMV_F32) return launch_fwd<float>(seg, nseg, gamma, beta, C, eps, 1, (hipStream_t)stream);
  LMV_FAIL(LMV_ERR_DTYPE, "layernorm_gelu: unsupported dtype %d", dtype);
}

extern "C" size_t lmv_layernorm_bwd_workspace_bytes(int64_t total_rows, int C, int dtype) {
  if (total_rows <= 0 || C <= 0) return 0;
  return (size_t)2048 * 2 * C * sizeof(float);     // upper bound: at most 2048 workgroups
}

extern "C" int lmv_layernorm_bwd(const lmv_ln_segment* seg, int nseg, const float* gamma, float* dgamma, float* dbeta, int C,
                                 void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (!seg) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: null segments");
  if (C <= 0 || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: C=%d must be a positive multiple of 8", C);
  if (!gamma || !dgamma || !dbeta) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: null affine / gradient buffer");
  if (dtype == LMV_BF16) return launch_bwd<bf16_t, false>(seg, nseg, gamma, nullptr, dgamma, dbeta, C, workspace, workspace_bytes, (hipStream_t)stream);
  if (dtype == LMV_F32) return launch_bwd<float, false>(seg, nseg, gamma, nullptr, dgamma, dbeta, C, workspace, workspace_bytes, (hipStream_t)stream);
  LMV_FAIL(LMV_ERR_DTYPE, "layernorm_bwd: unsupported dtype %d", dtype);
}

// The two halves of lmv_layernorm_bwd as separate calls: nothing inside a block consumes dgamma / dbeta, so a scheduler can leave the
// ~5 us reduce launch off the critical path (lmv_block_bwd enqueues it on the weight-gradient side stream).  Bit-identical results.
extern "C" int lmv_layernorm_bwd_partial(const lmv_ln_segment* seg, int nseg, const float* gamma, int C, void* workspace, size_t workspace_bytes,
                                         int* partial_rows, int dtype, void* stream) {
  if (!seg || !partial_rows) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd_partial: null segments / partial_rows");
  if (C <= 0 || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd_partial: C=%d must be a positive multiple of 8", C);
  if (!gamma) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd_partial: null affine");
  if (dtype == LMV_BF16) return launch_bwd<bf16_t, false>(seg, nseg, gamma, nullptr, nullptr, nullptr, C, workspace, workspace_bytes, (hipStream_t)stream, partial_rows);
  if (dtype == LMV_F32) return launch_bwd<float, false>(seg, nseg, gamma, nullptr, nullptr, nullptr, C, workspace, workspace_bytes, (hipStream_t)stream, partial_rows);
  LMV_FAIL(LMV_ERR_DTYPE, "layernorm_bwd_partial: unsupported dtype %d", dtype);
}
extern "C" int lmv_layernorm_bwd_reduce(const void* workspace, int partial_rows, int C, float* dgamma, float* dbeta, void* stream) {
  if (!workspace || partial_rows <= 0 || C <= 0 || (C % 8) || !dgamma || !dbeta) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd_reduce: bad arguments");
  return lmv_launch_partial_reduce(reinterpret_cast<const float*>(workspace), partial_rows, 2 * C, dgamma, C, dbeta, 0, (hipStream_t)stream);
}

extern "C" int lmv_layernorm_gelu_bwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, float* dgamma, float* dbeta, int C,
                                      void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (!seg) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_gelu_bwd: null segments");
  if (C <= 0 || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_gelu_bwd: C=%d must be a positive multiple of 8", C);
  if (!gamma || !beta || !dgamma || !dbeta) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_gelu_bwd: null affine / gradient buffer");
  if (dtype == LMV_BF16) return launch_bwd<bf16_t, true>(seg, nseg, gamma, beta, dgamma, dbeta, C, workspace, workspace_bytes, (hipStream_t)stream);
  if (dtype == LMV_F32) return launch_bwd<float, true>(seg, nseg, gamma, beta, dgamma, dbeta, C, workspace, workspace_bytes, (hipStream_t)stream);
  LMV_FAIL(LMV_ERR_DTYPE, "layernorm_gelu_bwd: unsupported dtype %d", dtype);
}
