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
// Backward: dgamma / dbeta are reduced registers -> wavefront shuffles -> LDS -> one partial row per
// workgroup; a second tiny kernel sums the partial rows (no global atomics: they serialise across XCDs).
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int BWD_MAXC = 2048;

template <typename T>
struct Segs {
  const T* x[2]; T* y[2]; float* stats[2]; const T* dy[2]; const T* dres[2]; T* dx[2];
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
                                                    int C, int lpr_log2, float eps) {
  constexpr int EPC = DT<T>::EPC;
  const int lpr = 1 << lpr_log2, nch = C / EPC;
  const int lir = threadIdx.x & (lpr - 1), rib = threadIdx.x >> lpr_log2, rpb = TPB >> lpr_log2;
  const float invC = 1.f / (float)C;
  for (int64_t row = (int64_t)blockIdx.x * rpb + rib; row < sg.total; row += (int64_t)gridDim.x * rpb) {
    const int s = row >= sg.rows0;
    const int64_t lr = row - (s ? sg.rows0 : 0);
    const T* xr = (s ? sg.x[1] : sg.x[0]) + lr * C;
    float v[NIT][EPC];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ch = lir + it * lpr;
      if (ch < nch) {
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(xr + ch * EPC), v[it]);
#pragma unroll
        for (int e = 0; e < EPC; ++e) sum += v[it][e];
      }
    }
    const float mean = group_sum(sum, lpr) * invC;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (lir + it * lpr < nch) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { const float d = v[it][e] - mean; q += d * d; }
      }
    }
    const float rstd = 1.f / sqrtf(group_sum(q, lpr) * invC + eps);
    T* yr = (s ? sg.y[1] : sg.y[0]) + lr * C;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ch = lir + it * lpr;
      if (ch < nch) {
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const float4 g4 = *reinterpret_cast<const float4*>(gamma + ch * EPC + e);
          const float4 b4 = *reinterpret_cast<const float4*>(beta + ch * EPC + e);
          o[e + 0] = (v[it][e + 0] - mean) * rstd * g4.x + b4.x;
          o[e + 1] = (v[it][e + 1] - mean) * rstd * g4.y + b4.y;
          o[e + 2] = (v[it][e + 2] - mean) * rstd * g4.z + b4.z;
          o[e + 3] = (v[it][e + 3] - mean) * rstd * g4.w + b4.w;
        }
        *reinterpret_cast<uint4*>(yr + ch * EPC) = f_to_chunk<T>(o);
      }
    }
    float* st = s ? sg.stats[1] : sg.stats[0];
    if (st && lir == 0) { st[lr * 2] = mean; st[lr * 2 + 1] = rstd; }
  }
}

template <typename T, int NIT>
__global__ __launch_bounds__(TPB) void ln_bwd_kernel(const Segs<T> sg, const float* __restrict__ gamma, float* __restrict__ partial, int C, int lpr_log2) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float s_dg[BWD_MAXC], s_db[BWD_MAXC];
  const int lpr = 1 << lpr_log2, nch = C / EPC;
  const int lir = threadIdx.x & (lpr - 1), rib = threadIdx.x >> lpr_log2, rpb = TPB >> lpr_log2;
  const float invC = 1.f / (float)C;
  for (int c = threadIdx.x; c < C; c += TPB) { s_dg[c] = 0.f; s_db[c] = 0.f; }
  float adg[NIT][EPC], adb[NIT][EPC], gm[NIT][EPC];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lir + it * lpr;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { adg[it][e] = 0.f; adb[it][e] = 0.f; gm[it][e] = (ch < nch) ? gamma[ch * EPC + e] : 0.f; }
  }
  for (int64_t row = (int64_t)blockIdx.x * rpb + rib; row < sg.total; row += (int64_t)gridDim.x * rpb) {
    const int s = row >= sg.rows0;
    const int64_t lr = row - (s ? sg.rows0 : 0);
    const float* st = s ? sg.stats[1] : sg.stats[0];
    const T* xr = (s ? sg.x[1] : sg.x[0]) + lr * C;
    const T* dyr = (s ? sg.dy[1] : sg.dy[0]) + lr * C;
    const T* drp = s ? sg.dres[1] : sg.dres[0];
    T* dxr = (s ? sg.dx[1] : sg.dx[0]) + lr * C;
    const float mean = st[lr * 2], rstd = st[lr * 2 + 1];
    float xh[NIT][EPC], g[NIT][EPC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ch = lir + it * lpr;
      if (ch < nch) {
        float xv[EPC], dv[EPC];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(xr + ch * EPC), xv);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(dyr + ch * EPC), dv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          xh[it][e] = (xv[e] - mean) * rstd;
          g[it][e] = dv[e] * gm[it][e];
          s1 += g[it][e]; s2 += g[it][e] * xh[it][e];
          adg[it][e] += dv[e] * xh[it][e]; adb[it][e] += dv[e];
        }
      }
    }
    s1 = group_sum(s1, lpr) * invC; s2 = group_sum(s2, lpr) * invC;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ch = lir + it * lpr;
      if (ch < nch) {
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = rstd * (g[it][e] - s1 - xh[it][e] * s2);
        if (drp) {
          float r[EPC];
          chunk_to_f<T>(*reinterpret_cast<const uint4*>(drp + lr * C + ch * EPC), r);
#pragma unroll
          for (int e = 0; e < EPC; ++e) o[e] += r[e];
        }
        *reinterpret_cast<uint4*>(dxr + ch * EPC) = f_to_chunk<T>(o);
      }
    }
  }
  __syncthreads();
  // column sums: first across the row groups of this wavefront (lanes with equal `lir`), then one LDS add per wave
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lir + it * lpr;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float a = adg[it][e], b = adb[it][e];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
        if (o >= lpr) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if ((threadIdx.x & 63) < lpr && ch < nch) { atomicAdd(&s_dg[ch * EPC + e], a); atomicAdd(&s_db[ch * EPC + e], b); }
    }
  }
  __syncthreads();
  float* prow = partial + (int64_t)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < C; c += TPB) { prow[c] = s_dg[c]; prow[C + c] = s_db[c]; }
}

// lanes per row (log2) and iterations for a row of `nch` 16-byte chunks
inline bool pick_geometry(int nch, int max_it, int* lpr_log2, int* nit) {
  int best = -1, best_score = 1 << 30, best_it = 0;
  for (int l = 2; l <= 6; ++l) {
    const int lpr = 1 << l, it = (nch + lpr - 1) / lpr;
    if (it > max_it) continue;
    const int waste = it * lpr - nch;                            // idle lane-iterations per row
    const int score = waste * 100 + (it > 3 ? it - 3 : 3 - it);  // then ~3 loads in flight per lane
    if (score < best_score) { best = l; best_score = score; best_it = it; }
  }
  if (best < 0) return false;
  *lpr_log2 = best; *nit = best_it;
  return true;
}

inline int bwd_blocks(int64_t rows, int l2) {
  const int rpb = TPB >> l2;
  int64_t blocks = (rows + 2 * rpb - 1) / (2 * rpb);           // >= 2 rows per row group
  return (int)(blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks));
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
  }
  sg->rows0 = seg[0].rows;
  sg->total = seg[0].rows + (nseg == 2 ? seg[1].rows : 0);
  return LMV_OK;
}

template <typename T>
int launch_fwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, int C, float eps, hipStream_t st) {
  Segs<T> sg;
  if (int rc = fill(&sg, seg, nseg, false, "layernorm")) return rc;
  int l2, nit;
  if (!pick_geometry(C / DT<T>::EPC, sizeof(T) == 4 ? 8 : 6, &l2, &nit)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm: C=%d too wide", C);
  const int rpb = TPB >> l2;
  int64_t blocks = (sg.total + rpb - 1) / rpb; if (blocks > 4096) blocks = 4096;
  dim3 grid((int)blocks), block(TPB);
#define LN_FWD_CASE(N) case N: hipLaunchKernelGGL((ln_fwd_kernel<T, N>), grid, block, 0, st, sg, gamma, beta, C, l2, eps); break;
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

template <typename T>
int launch_bwd(const lmv_ln_segment* seg, int nseg, const float* gamma, float* dgamma, float* dbeta, int C, void* ws, size_t ws_bytes, hipStream_t st) {
  Segs<T> sg;
  if (int rc = fill(&sg, seg, nseg, true, "layernorm_bwd")) return rc;
  int l2, nit;
  if (C > BWD_MAXC || !pick_geometry(C / DT<T>::EPC, sizeof(T) == 4 ? 8 : 6, &l2, &nit)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: C=%d too wide", C);
  const int blocks = bwd_blocks(sg.total, l2);
  const size_t need = (size_t)blocks * 2 * C * sizeof(float);
  if (!ws || ws_bytes < need) LMV_FAIL(LMV_ERR_WORKSPACE, "layernorm_bwd: workspace %zu < %zu bytes", ws_bytes, need);
  float* partial = reinterpret_cast<float*>(ws);
  dim3 grid(blocks), block(TPB);
#define LN_BWD_CASE(N) case N: hipLaunchKernelGGL((ln_bwd_kernel<T, N>), grid, block, 0, st, sg, gamma, partial, C, l2); break;
  switch (nit) {
    LN_BWD_CASE(1) LN_BWD_CASE(2) LN_BWD_CASE(3) LN_BWD_CASE(4) LN_BWD_CASE(5) LN_BWD_CASE(6)
    default:
      if constexpr (sizeof(T) == 4) { switch (nit) { LN_BWD_CASE(7) LN_BWD_CASE(8) } }
      else LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: C=%d too wide", C);
  }
#undef LN_BWD_CASE
  LMV_CHECK_LAUNCH("layernorm_bwd");
  return lmv_launch_partial_reduce(partial, blocks, 2 * C, dgamma, C, dbeta, 0, st);
}

}  // namespace

extern "C" int lmv_layernorm_fwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, int C, float eps, int dtype, void* stream) {
  if (!seg) LMV_FAIL(LMV_ERR_SHAPE, "layernorm: null segments");
  if (C <= 0 || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm: C=%d must be a positive multiple of 8", C);
  if (!gamma || !beta || !lmv_aligned16(gamma) || !lmv_aligned16(beta)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm: null or misaligned affine");
  if (dtype == LMV_BF16) return launch_fwd<bf16_t>(seg, nseg, gamma, beta, C, eps, (hipStream_t)stream);
  if (dtype == LMV_F32) return launch_fwd<float>(seg, nseg, gamma, beta, C, eps, (hipStream_t)stream);
  LMV_FAIL(LMV_ERR_DTYPE, "layernorm: unsupported dtype %d", dtype);
}

extern "C" size_t lmv_layernorm_bwd_workspace_bytes(int64_t total_rows, int C, int dtype) {
  if (total_rows <= 0 || C <= 0) return 0;
  return (size_t)512 * 2 * C * sizeof(float);     // upper bound: at most 512 workgroups
}

extern "C" int lmv_layernorm_bwd(const lmv_ln_segment* seg, int nseg, const float* gamma, float* dgamma, float* dbeta, int C,
                                 void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (!seg) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: null segments");
  if (C <= 0 || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: C=%d must be a positive multiple of 8", C);
  if (!gamma || !dgamma || !dbeta) LMV_FAIL(LMV_ERR_SHAPE, "layernorm_bwd: null affine / gradient buffer");
  if (dtype == LMV_BF16) return launch_bwd<bf16_t>(seg, nseg, gamma, dgamma, dbeta, C, workspace, workspace_bytes, (hipStream_t)stream);
  if (dtype == LMV_F32) return launch_bwd<float>(seg, nseg, gamma, dgamma, dbeta, C, workspace, workspace_bytes, (hipStream_t)stream);
  LMV_FAIL(LMV_ERR_DTYPE, "layernorm_bwd: unsupported dtype %d", dtype);
}
