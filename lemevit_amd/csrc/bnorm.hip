// bnorm.hip -- training-mode BatchNorm2d (+ exact GELU) over channels-last activations (memory-bound; graded on HBM GB/s).
//
// The stem and the stage transitions are Conv -> BatchNorm (-> GELU) (models/lemevit.py:698-704, 714-717) and the tail
// is BatchNorm2d over the last feature map (:773, 822).  In token-major / NHWC memory a feature map is a [rows = B*H*W][C]
// matrix and BatchNorm is a COLUMN statistic, so the kernels share one thread layout: a workgroup's 256 threads are
// rows_per_iter x (C / EPC) 16-byte chunks, every thread keeps ONE channel chunk for its whole life (per-channel constants
// live in registers) and walks rows  r0 + slot, r0 + slot + rows_per_iter, ...  with unconditional 16-byte loads.
//
//   forward : bn_stats (per-workgroup partial sums of x and x^2, fp32) -> bn_finalize (sums the partial rows in fp64:
//             mean, 1/sqrt(var + eps), running-stat update with the unbiased variance, as torch) -> bn_apply
//             (y = x * a + b per channel, optional GELU).  Two reads and one write of the feature map.
//   backward: bn_bwd_stats (sum dy', sum dy' * xhat;  dy' = dy * GELU'(u) with u recomputed from x) -> bn_bwd_finalize
//             (dgamma, dbeta) -> bn_bwd_apply (dx = gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' * xhat))).
// No atomics: partial rows + a fixed-order reduction, so results are run-to-run reproducible.
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAX_BLOCKS = 1024;

struct Geo { int64_t rows; int C, nch, rpi, blocks; int64_t rows_per_block; };

template <typename T>
inline Geo make_geo(int64_t rows, int C) {
  Geo g;
  g.rows = rows; g.C = C; g.nch = C / DT<T>::EPC;
  g.rpi = TPB / g.nch;                                         // rows in flight per workgroup iteration
  int64_t blocks = (rows + g.rpi * 8 - 1) / (g.rpi * 8);      // >= 8 rows per thread
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  g.rows_per_block = (rows + blocks - 1) / blocks;
  g.blocks = (int)((rows + g.rows_per_block - 1) / g.rows_per_block);
  return g;
}

// reduce the rpi per-slot vectors of a workgroup (one [nq][EPC] value set per thread) to one partial row [nq][C]
template <int NQ, int EPC>
__device__ __forceinline__ void block_reduce_store(float (&v)[NQ][EPC], float* red, float* prow, int C, int nch, int rpi, int slot, int ch) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    __syncthreads();
    if (slot < rpi) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) red[slot * C + ch * EPC + e] = v[q][e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += TPB) {
      float s = 0.f;
      for (int k = 0; k < rpi; ++k) s += red[k * C + c];
      prow[q * C + c] = s;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(TPB) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ partial, const Geo g) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[TPB * EPC];
  const int ch = threadIdx.x % g.nch, slot = threadIdx.x / g.nch;
  float acc[2][EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block, r1 = min(g.rows, r0 + g.rows_per_block);
  if (slot < g.rpi) {
    const T* p = x + ch * EPC;
    int64_t r = r0 + slot;
    for (; r + 3 * g.rpi < r1; r += 4 * g.rpi) {               // four independent 16-byte loads in flight
      uint4 raw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) raw[j] = *reinterpret_cast<const uint4*>(p + (r + (int64_t)j * g.rpi) * g.C);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[EPC];
        chunk_to_f<T>(raw[j], v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { acc[0][e] += v[e]; acc[1][e] = fmaf(v[e], v[e], acc[1][e]); }
      }
    }
    for (; r < r1; r += g.rpi) {
      float v[EPC];
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(p + r * g.C), v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { acc[0][e] += v[e]; acc[1][e] = fmaf(v[e], v[e], acc[1][e]); }
    }
  }
  block_reduce_store<2, EPC>(acc, red, partial + (int64_t)blockIdx.x * 2 * g.C, g.C, g.nch, g.rpi, slot, ch);
}

// column sums of the partial rows in fp64: a workgroup owns 8 channels x 32 row groups (thread (cx, rg) walks partial rows
// rg, rg + 32, ...), combined through LDS in a fixed order
__device__ __forceinline__ bool column_sums(const float* __restrict__ partial, int nblocks, int C, double* s_out, double* q_out, int* c_out) {
  __shared__ double red[2][32][8];
  const int cx = threadIdx.x & 7, rg = threadIdx.x >> 3, c = blockIdx.x * 8 + cx;
  double s = 0.0, q = 0.0;
  if (c < C) {
    int b = rg;
    for (; b + 7 * 32 < nblocks; b += 8 * 32) {                 // 16 independent loads in flight, summed in row order
      float vs[8], vq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { vs[j] = partial[(int64_t)(b + j * 32) * 2 * C + c]; vq[j] = partial[(int64_t)(b + j * 32) * 2 * C + C + c]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { s += (double)vs[j]; q += (double)vq[j]; }
    }
    for (; b < nblocks; b += 32) { s += (double)partial[(int64_t)b * 2 * C + c]; q += (double)partial[(int64_t)b * 2 * C + C + c]; }
  }
  red[0][rg][cx] = s; red[1][rg][cx] = q;
  __syncthreads();
  if (rg != 0 || c >= C) return false;
  for (int k = 1; k < 32; ++k) { s += red[0][k][cx]; q += red[1][k][cx]; }
  *s_out = s; *q_out = q; *c_out = c;
  return true;
}

// stats[c] = mean, stats[C + c] = rstd; running statistics as torch.nn.BatchNorm2d (momentum, unbiased variance)
__global__ __launch_bounds__(TPB) void bn_finalize_kernel(const float* __restrict__ partial, int nblocks, int C, double inv_rows, double unbias,
                                                         float eps, float momentum, float* __restrict__ stats, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var) {
  double s, q; int c;
  if (!column_sums(partial, nblocks, C, &s, &q, &c)) return;
  const double mean = s * inv_rows;
  double var = q * inv_rows - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
}

template <typename T, bool ACT>
__global__ __launch_bounds__(TPB) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, T* __restrict__ y, const Geo g) {
  constexpr int EPC = DT<T>::EPC;
  const int ch = threadIdx.x % g.nch, slot = threadIdx.x / g.nch;
  if (slot >= g.rpi) return;
  float a[EPC], b[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = ch * EPC + e;
    a[e] = stats[g.C + c] * gamma[c];
    b[e] = beta[c] - stats[c] * a[e];
  }
  const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block, r1 = min(g.rows, r0 + g.rows_per_block);
  auto one = [&](const uint4& raw, int64_t r) {
    float v[EPC];
    chunk_to_f<T>(raw, v);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      v[e] = fmaf(v[e], a[e], b[e]);
      if (ACT) v[e] = sizeof(T) == 4 ? gelu_f(v[e]) : gelu_fast_f(v[e]);
    }
    *reinterpret_cast<uint4*>(y + r * g.C + ch * EPC) = f_to_chunk<T>(v);
  };
  const T* p = x + ch * EPC;
  int64_t r = r0 + slot;
  for (; r + 3 * g.rpi < r1; r += 4 * g.rpi) {
    uint4 raw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = *reinterpret_cast<const uint4*>(p + (r + (int64_t)j * g.rpi) * g.C);
#pragma unroll
    for (int j = 0; j < 4; ++j) one(raw[j], r + (int64_t)j * g.rpi);
  }
  for (; r < r1; r += g.rpi) one(*reinterpret_cast<const uint4*>(p + r * g.C), r);
}

// dy' (the gradient at the BatchNorm output) and xhat of one chunk
template <typename T, bool ACT, int EPC>
__device__ __forceinline__ void bwd_terms(const uint4& rx, const uint4& rdy, const float (&mean)[EPC], const float (&rstd)[EPC],
                                          const float (&gm)[EPC], const float (&bt)[EPC], float (&dyp)[EPC], float (&xh)[EPC]) {
  float xv[EPC];
  chunk_to_f<T>(rx, xv);
  chunk_to_f<T>(rdy, dyp);
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    xh[e] = (xv[e] - mean[e]) * rstd[e];
    if (ACT) {
      const float u = fmaf(xh[e], gm[e], bt[e]);
      dyp[e] *= sizeof(T) == 4 ? gelu_grad_f(u) : gelu_grad_fast_f(u);
    }
  }
}

template <typename T, bool ACT>
__global__ __launch_bounds__(TPB) void bn_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ partial, const Geo g) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[TPB * EPC];
  const int ch = threadIdx.x % g.nch, slot = threadIdx.x / g.nch;
  float acc[2][EPC], mean[EPC], rstd[EPC], gm[EPC], bt[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = min(ch * EPC + e, g.C - 1);
    acc[0][e] = 0.f; acc[1][e] = 0.f; mean[e] = stats[c]; rstd[e] = stats[g.C + c]; gm[e] = gamma[c]; bt[e] = beta[c];
  }
  const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block, r1 = min(g.rows, r0 + g.rows_per_block);
  if (slot < g.rpi) {
    const int64_t co = ch * EPC;
    int64_t r = r0 + slot;
    for (; r + g.rpi < r1; r += 2 * g.rpi) {                   // two rows = four 16-byte loads in flight
      const uint4 x0 = *reinterpret_cast<const uint4*>(x + r * g.C + co), d0 = *reinterpret_cast<const uint4*>(dy + r * g.C + co);
      const uint4 x1 = *reinterpret_cast<const uint4*>(x + (r + g.rpi) * g.C + co), d1 = *reinterpret_cast<const uint4*>(dy + (r + g.rpi) * g.C + co);
      float dyp[EPC], xh[EPC];
      bwd_terms<T, ACT, EPC>(x0, d0, mean, rstd, gm, bt, dyp, xh);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { acc[0][e] += dyp[e]; acc[1][e] = fmaf(dyp[e], xh[e], acc[1][e]); }
      bwd_terms<T, ACT, EPC>(x1, d1, mean, rstd, gm, bt, dyp, xh);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { acc[0][e] += dyp[e]; acc[1][e] = fmaf(dyp[e], xh[e], acc[1][e]); }
    }
    for (; r < r1; r += g.rpi) {
      float dyp[EPC], xh[EPC];
      bwd_terms<T, ACT, EPC>(*reinterpret_cast<const uint4*>(x + r * g.C + co), *reinterpret_cast<const uint4*>(dy + r * g.C + co), mean, rstd, gm, bt, dyp, xh);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { acc[0][e] += dyp[e]; acc[1][e] = fmaf(dyp[e], xh[e], acc[1][e]); }
    }
  }
  block_reduce_store<2, EPC>(acc, red, partial + (int64_t)blockIdx.x * 2 * g.C, g.C, g.nch, g.rpi, slot, ch);
}

// sums[c] = sum dy' (= dbeta), sums[C + c] = sum dy' * xhat (= dgamma)
__global__ __launch_bounds__(TPB) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int C, float* __restrict__ sums,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  double s, q; int c;
  if (!column_sums(partial, nblocks, C, &s, &q, &c)) return;
  sums[c] = (float)s; sums[C + c] = (float)q;
  dbeta[c] = (float)s; dgamma[c] = (float)q;
}

template <typename T, bool ACT>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ sums, float inv_rows, T* __restrict__ dx, const Geo g) {
  constexpr int EPC = DT<T>::EPC;
  const int ch = threadIdx.x % g.nch, slot = threadIdx.x / g.nch;
  if (slot >= g.rpi) return;
  float mean[EPC], rstd[EPC], gm[EPC], bt[EPC], k[EPC], m1[EPC], m2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = ch * EPC + e;
    mean[e] = stats[c]; rstd[e] = stats[g.C + c]; gm[e] = gamma[c]; bt[e] = beta[c];
    k[e] = gm[e] * rstd[e]; m1[e] = sums[c] * inv_rows; m2[e] = sums[g.C + c] * inv_rows;
  }
  const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block, r1 = min(g.rows, r0 + g.rows_per_block);
  const int64_t co = ch * EPC;
  auto one = [&](const uint4& rx, const uint4& rd, int64_t r) {
    float dyp[EPC], xh[EPC], o[EPC];
    bwd_terms<T, ACT, EPC>(rx, rd, mean, rstd, gm, bt, dyp, xh);
#pragma unroll
    for (int e = 0; e < EPC; ++e) o[e] = k[e] * (dyp[e] - m1[e] - xh[e] * m2[e]);
    *reinterpret_cast<uint4*>(dx + r * g.C + co) = f_to_chunk<T>(o);
  };
  int64_t r = r0 + slot;
  for (; r + g.rpi < r1; r += 2 * g.rpi) {
    const uint4 x0 = *reinterpret_cast<const uint4*>(x + r * g.C + co), d0 = *reinterpret_cast<const uint4*>(dy + r * g.C + co);
    const uint4 x1 = *reinterpret_cast<const uint4*>(x + (r + g.rpi) * g.C + co), d1 = *reinterpret_cast<const uint4*>(dy + (r + g.rpi) * g.C + co);
    one(x0, d0, r);
    one(x1, d1, r + g.rpi);
  }
  for (; r < r1; r += g.rpi) one(*reinterpret_cast<const uint4*>(x + r * g.C + co), *reinterpret_cast<const uint4*>(dy + r * g.C + co), r);
}

inline int check(const char* name, const void* a, const void* b, int64_t rows, int C, int dtype) {
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "%s: unsupported dtype %d", name, dtype);
  const int epc = dtype == LMV_BF16 ? 8 : 4;
  if (rows <= 1 || C <= 0 || (C % epc) || C / epc > TPB || rows * (int64_t)C >= ((int64_t)1 << 40))
    LMV_FAIL(LMV_ERR_SHAPE, "%s: bad shape rows=%lld C=%d (rows > 1, C %% %d == 0, C <= %d)", name, (long long)rows, C, epc, TPB * epc);
  if (!a || !b || !lmv_aligned16(a) || !lmv_aligned16(b)) LMV_FAIL(LMV_ERR_SHAPE, "%s: null or misaligned operand", name);
  return LMV_OK;
}

template <typename T>
int fwd_t(const T* x, const float* gamma, const float* beta, float* rm, float* rv, float momentum, float eps, int act, T* y, float* stats,
          int64_t rows, int C, float* ws, hipStream_t st) {
  const Geo g = make_geo<T>(rows, C);
  hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(g.blocks), dim3(TPB), 0, st, x, ws, g);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(TPB), 0, st, (const float*)ws, g.blocks, C, 1.0 / (double)rows,
                     (double)rows / (double)(rows - 1), eps, momentum, stats, rm, rv);
  if (act) hipLaunchKernelGGL((bn_apply_kernel<T, true>), dim3(g.blocks), dim3(TPB), 0, st, x, (const float*)stats, gamma, beta, y, g);
  else hipLaunchKernelGGL((bn_apply_kernel<T, false>), dim3(g.blocks), dim3(TPB), 0, st, x, (const float*)stats, gamma, beta, y, g);
  LMV_CHECK_LAUNCH("batchnorm_train_fwd");
  return LMV_OK;
}

template <typename T>
int bwd_t(const T* dy, const T* x, const float* gamma, const float* beta, const float* stats, int act, T* dx, float* dgamma, float* dbeta,
          int64_t rows, int C, float* ws, hipStream_t st) {
  const Geo g = make_geo<T>(rows, C);
  float* sums = ws + (size_t)MAX_BLOCKS * 2 * C;
  if (act) hipLaunchKernelGGL((bn_bwd_stats_kernel<T, true>), dim3(g.blocks), dim3(TPB), 0, st, dy, x, stats, gamma, beta, ws, g);
  else hipLaunchKernelGGL((bn_bwd_stats_kernel<T, false>), dim3(g.blocks), dim3(TPB), 0, st, dy, x, stats, gamma, beta, ws, g);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(TPB), 0, st, (const float*)ws, g.blocks, C, sums, dgamma, dbeta);
  if (act) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(g.blocks), dim3(TPB), 0, st, dy, x, stats, gamma, beta, (const float*)sums, 1.f / (float)rows, dx, g);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), dim3(g.blocks), dim3(TPB), 0, st, dy, x, stats, gamma, beta, (const float*)sums, 1.f / (float)rows, dx, g);
  LMV_CHECK_LAUNCH("batchnorm_train_bwd");
  return LMV_OK;
}

}  // namespace

extern "C" size_t lmv_batchnorm_workspace_bytes(int C) { return C > 0 ? ((size_t)MAX_BLOCKS * 2 * C + 2 * (size_t)C) * sizeof(float) : 0; }

extern "C" int lmv_batchnorm_train_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                       float eps, int act, void* y, float* stats, int64_t rows, int C, void* workspace, size_t workspace_bytes,
                                       int dtype, void* stream) {
  if (int rc = check("batchnorm_train_fwd", x, y, rows, C, dtype)) return rc;
  if (!gamma || !beta || !stats) LMV_FAIL(LMV_ERR_SHAPE, "batchnorm_train_fwd: null gamma / beta / stats");
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU) LMV_FAIL(LMV_ERR_SHAPE, "batchnorm_train_fwd: act must be NONE or GELU");
  if (!workspace || workspace_bytes < lmv_batchnorm_workspace_bytes(C)) LMV_FAIL(LMV_ERR_WORKSPACE, "batchnorm_train_fwd: workspace %zu < %zu bytes", workspace_bytes, lmv_batchnorm_workspace_bytes(C));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMV_BF16) return fwd_t<bf16_t>((const bf16_t*)x, gamma, beta, running_mean, running_var, momentum, eps, act, (bf16_t*)y, stats, rows, C, (float*)workspace, st);
  return fwd_t<float>((const float*)x, gamma, beta, running_mean, running_var, momentum, eps, act, (float*)y, stats, rows, C, (float*)workspace, st);
}

extern "C" int lmv_batchnorm_train_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats, int act, void* dx,
                                       float* dgamma, float* dbeta, int64_t rows, int C, void* workspace, size_t workspace_bytes, int dtype,
                                       void* stream) {
  if (int rc = check("batchnorm_train_bwd", dy, x, rows, C, dtype)) return rc;
  if (!gamma || !beta || !stats || !dx || !dgamma || !dbeta || !lmv_aligned16(dx)) LMV_FAIL(LMV_ERR_SHAPE, "batchnorm_train_bwd: null or misaligned operand");
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU) LMV_FAIL(LMV_ERR_SHAPE, "batchnorm_train_bwd: act must be NONE or GELU");
  if (!workspace || workspace_bytes < lmv_batchnorm_workspace_bytes(C)) LMV_FAIL(LMV_ERR_WORKSPACE, "batchnorm_train_bwd: workspace %zu < %zu bytes", workspace_bytes, lmv_batchnorm_workspace_bytes(C));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMV_BF16) return bwd_t<bf16_t>((const bf16_t*)dy, (const bf16_t*)x, gamma, beta, stats, act, (bf16_t*)dx, dgamma, dbeta, rows, C, (float*)workspace, st);
  return bwd_t<float>((const float*)dy, (const float*)x, gamma, beta, stats, act, (float*)dx, dgamma, dbeta, rows, C, (float*)workspace, st);
}
