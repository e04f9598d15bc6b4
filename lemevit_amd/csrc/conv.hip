// conv.hip -- conditional position embedding of a LeMeBlock: y = x + dwconv3x3(x) + bias on a
// token-major (NHWC) tensor (models/lemevit.py:510,546), plus its two backward kernels.
// Memory-bound: every thread owns one pixel x 16 bytes of channels; the 9 taps hit L1/L2.
// Depth-wise weights are staged once per workgroup into LDS as [tap][C] fp32.
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXC = 1024;

// FLIP = false: y = x + bias + sum_t w[c][t] * x[h+dy-1][w+dx-1]
// FLIP = true : y = x +        sum_t w[c][8-t] * x[h+dy-1][w+dx-1]   (transpose conv = backward-data)
template <typename T, bool FLIP>
__global__ __launch_bounds__(TPB) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                    T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float sw[9 * MAXC];
  for (int i = threadIdx.x; i < 9 * C; i += TPB) {
    const int c = i / 9, t = i % 9;
    sw[(FLIP ? 8 - t : t) * C + c] = wgt[i];
  }
  __syncthreads();
  const int nch = C / EPC;
  const int64_t total = (int64_t)B * H * W * nch;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int ch = (int)(i % nch);
    const int64_t pix = i / nch;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const int c0 = ch * EPC;
    float acc[EPC];
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + pix * C + c0), acc);
    if (!FLIP && bias) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += bias[c0 + e];
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int hh = h + dy - 1;
      if (hh < 0 || hh >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ww = w + dx - 1;
        if (ww < 0 || ww >= W) continue;
        float v[EPC];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + (pix + (int64_t)(dy - 1) * W + (dx - 1)) * C + c0), v);
        const float* wt = sw + (dy * 3 + dx) * C + c0;
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += wt[e] * v[e];
      }
    }
    *reinterpret_cast<uint4*>(y + pix * C + c0) = f_to_chunk<T>(acc);
  }
}

// dw[c][t] += sum_pix dy[pix][c] * x[pix + tap t][c] ; db[c] += sum_pix dy[pix][c]
// blockDim = nch * slots: thread (slot, chunk) walks pixels slot, slot + stride, ...
template <typename T>
__global__ __launch_bounds__(TPB) void dwconv_bwd_w_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw,
                                                          float* __restrict__ db, int B, int H, int W, int C, int slots) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float sacc[10 * MAXC];
  const int nch = C / EPC;
  for (int i = threadIdx.x; i < 10 * C; i += blockDim.x) sacc[i] = 0.f;
  __syncthreads();
  const int ch = threadIdx.x % nch, slot = threadIdx.x / nch, c0 = ch * EPC;
  float a[10][EPC];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[t][e] = 0.f;
  const int64_t npix = (int64_t)B * H * W;
  if (slot < slots) {
    for (int64_t pix = (int64_t)blockIdx.x * slots + slot; pix < npix; pix += (int64_t)gridDim.x * slots) {
      const int w = (int)(pix % W), h = (int)((pix / W) % H);
      float g[EPC];
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(dy + pix * C + c0), g);
#pragma unroll
      for (int e = 0; e < EPC; ++e) a[9][e] += g[e];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int hh = h + ky - 1;
        if (hh < 0 || hh >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ww = w + kx - 1;
          if (ww < 0 || ww >= W) continue;
          float v[EPC];
          chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + (pix + (int64_t)(ky - 1) * W + (kx - 1)) * C + c0), v);
#pragma unroll
          for (int e = 0; e < EPC; ++e) a[ky * 3 + kx][e] += g[e] * v[e];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
      for (int e = 0; e < EPC; ++e) atomicAdd(&sacc[t * C + c0 + e], a[t][e]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) {
    const int t = i / C, c = i % C;
    atomicAdd(dw + c * 9 + t, sacc[i]);
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(db + c, sacc[9 * C + c]);
}

inline int check(const char* name, const void* a, const void* b, int B, int H, int W, int C, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || C > MAXC) LMV_FAIL(LMV_ERR_SHAPE, "%s: bad shape B=%d H=%d W=%d C=%d (C %% 8 == 0, C <= %d)", name, B, H, W, C, MAXC);
  if (!a || !b || !lmv_aligned16(a) || !lmv_aligned16(b)) LMV_FAIL(LMV_ERR_SHAPE, "%s: null or misaligned operand", name);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "%s: unsupported dtype %d", name, dtype);
  return LMV_OK;
}

inline int grid_items(int64_t items) {
  int64_t b = (items + TPB - 1) / TPB;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int lmv_dwconv3x3_residual_fwd(const void* x, const float* weight, const float* bias, void* y, int B, int H, int W, int C,
                                          int dtype, void* stream) {
  if (int rc = check("dwconv_fwd", x, y, B, H, W, C, dtype)) return rc;
  if (!weight) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_fwd: null weight");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_kernel<bf16_t, false>), dim3(grid_items((int64_t)B * H * W * (C / 8))), dim3(TPB), 0, st, (const bf16_t*)x, weight, bias, (bf16_t*)y, B, H, W, C);
  else hipLaunchKernelGGL((dwconv_kernel<float, false>), dim3(grid_items((int64_t)B * H * W * (C / 4))), dim3(TPB), 0, st, (const float*)x, weight, bias, (float*)y, B, H, W, C);
  LMV_CHECK_LAUNCH("dwconv_fwd");
  return LMV_OK;
}

extern "C" int lmv_dwconv3x3_residual_bwd_data(const void* dy, const float* weight, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
  if (int rc = check("dwconv_bwd_data", dy, dx, B, H, W, C, dtype)) return rc;
  if (!weight) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_data: null weight");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_kernel<bf16_t, true>), dim3(grid_items((int64_t)B * H * W * (C / 8))), dim3(TPB), 0, st, (const bf16_t*)dy, weight, (const float*)nullptr, (bf16_t*)dx, B, H, W, C);
  else hipLaunchKernelGGL((dwconv_kernel<float, true>), dim3(grid_items((int64_t)B * H * W * (C / 4))), dim3(TPB), 0, st, (const float*)dy, weight, (const float*)nullptr, (float*)dx, B, H, W, C);
  LMV_CHECK_LAUNCH("dwconv_bwd_data");
  return LMV_OK;
}

extern "C" int lmv_dwconv3x3_bwd_weight(const void* dy, const void* x, float* dweight, float* dbias, int B, int H, int W, int C, int dtype, void* stream) {
  if (int rc = check("dwconv_bwd_weight", dy, x, B, H, W, C, dtype)) return rc;
  if (!dweight || !dbias) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_weight: null gradient buffer");
  hipStream_t st = (hipStream_t)stream;
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  if (nch > TPB) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_weight: C=%d too wide", C);
  const int slots = TPB / nch, threads = slots * nch;
  const int64_t npix = (int64_t)B * H * W;
  int64_t blocks = (npix + slots * 8 - 1) / (slots * 8);   // >= 8 pixels per thread
  if (blocks > 256) blocks = 256; if (blocks < 1) blocks = 1;
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_bwd_w_kernel<bf16_t>), dim3((int)blocks), dim3(threads), 0, st, (const bf16_t*)dy, (const bf16_t*)x, dweight, dbias, B, H, W, C, slots);
  else hipLaunchKernelGGL((dwconv_bwd_w_kernel<float>), dim3((int)blocks), dim3(threads), 0, st, (const float*)dy, (const float*)x, dweight, dbias, B, H, W, C, slots);
  LMV_CHECK_LAUNCH("dwconv_bwd_weight");
  return LMV_OK;
}
