// conv.hip -- conditional position embedding of a LeMeBlock: y = x + dwconv3x3(x) + bias on a
// token-major (NHWC) tensor (models/lemevit.py:510,546), plus its two backward kernels.
// Memory-bound: every thread owns one pixel x 16 bytes of channels; the 9 taps hit L1/L2.
// Grid = (pixel-row segments, H, B): no 64-bit index arithmetic in the kernels.
// Depth-wise weights are staged once per workgroup into LDS as [tap][C] fp32.
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXC = 1024;

// FLIP = false: y = x + bias + sum_t w[c][t] * x[h+dy-1][w+dx-1]
// FLIP = true : y = x +        sum_t w[c][8-t] * x[h+dy-1][w+dx-1]   (transpose conv = backward-data)
template <typename T, bool FLIP>
__global__ __launch_bounds__(TPB) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                    T* __restrict__ y, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float sw[9 * MAXC];
  for (int i = threadIdx.x; i < 9 * C; i += TPB) {
    const int c = i / 9, t = i - c * 9;
    sw[(FLIP ? 8 - t : t) * C + c] = wgt[i];
  }
  __syncthreads();
  const int nch = C / EPC;
  const int idx = blockIdx.x * TPB + threadIdx.x;      // (w, chunk) within image row h of sample b
  if (idx >= W * nch) return;
  const int w = idx / nch, c0 = (idx - w * nch) * EPC;
  const int h = blockIdx.y;
  const int64_t pix = ((int64_t)blockIdx.z * H + h) * W + w;
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
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + (pix + (dy - 1) * W + (dx - 1)) * C + c0), v);
      const float* wt = sw + (dy * 3 + dx) * C + c0;
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += wt[e] * v[e];
    }
  }
  *reinterpret_cast<uint4*>(y + pix * C + c0) = f_to_chunk<T>(acc);
}

// dw[c][t] += sum_pix dy[pix][c] * x[pix + tap t][c] ; db[c] += sum_pix dy[pix][c]
// blockDim = nch * slots.  Thread (slot, chunk) walks the pixels  p = slot + k * slots  of this workgroup's image rows
// and keeps 10 x EPC running sums; the `slots` partial sums of a channel are then combined through LDS (one tap at a
// time) and the workgroup writes ONE partial row [10][C] -- no atomics anywhere (partial_reduce_kernel finishes).
template <typename T>
__global__ __launch_bounds__(TPB) void dwconv_bwd_w_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ partial,
                                                          int H, int W, int C, int slots, int rows_per_block) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float sred[TPB * 8];
  const int nch = C / EPC;
  const int ch = threadIdx.x % nch, slot = threadIdx.x / nch, c0 = ch * EPC;
  float a[10][EPC];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[t][e] = 0.f;
  const int b = blockIdx.y;
  const int h0 = blockIdx.x * rows_per_block, h1 = min(H, h0 + rows_per_block);
  const int npix = (h1 - h0) * W;
  const int64_t base = ((int64_t)b * H + h0) * W;
  for (int p = slot; p < npix; p += slots) {
    const int hl = p / W, w = p - hl * W, h = h0 + hl;
    const int64_t pix = base + p;
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
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + (pix + (ky - 1) * W + (kx - 1)) * C + c0), v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) a[ky * 3 + kx][e] += g[e] * v[e];
      }
    }
  }
  float* prow = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 10 * C;
#pragma unroll 1
  for (int t = 0; t < 10; ++t) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPC; ++e) sred[slot * C + c0 + e] = a[t][e];       // [slots][C] (slots * C <= TPB * 8)
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float s = 0.f;
      for (int k = 0; k < slots; ++k) s += sred[k * C + c];
      prow[t * C + c] = s;
    }
  }
}

inline int check(const char* name, const void* a, const void* b, int B, int H, int W, int C, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || C > MAXC || B > 65535 || H > 65535)
    LMV_FAIL(LMV_ERR_SHAPE, "%s: bad shape B=%d H=%d W=%d C=%d (C %% 8 == 0, C <= %d)", name, B, H, W, C, MAXC);
  if (!a || !b || !lmv_aligned16(a) || !lmv_aligned16(b)) LMV_FAIL(LMV_ERR_SHAPE, "%s: null or misaligned operand", name);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "%s: unsupported dtype %d", name, dtype);
  return LMV_OK;
}

template <bool FLIP>
int launch_dwconv(const void* x, const float* weight, const float* bias, void* y, int B, int H, int W, int C, int dtype, hipStream_t st) {
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  dim3 grid((W * nch + TPB - 1) / TPB, H, B), block(TPB);
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_kernel<bf16_t, FLIP>), grid, block, 0, st, (const bf16_t*)x, weight, bias, (bf16_t*)y, H, W, C);
  else hipLaunchKernelGGL((dwconv_kernel<float, FLIP>), grid, block, 0, st, (const float*)x, weight, bias, (float*)y, H, W, C);
  LMV_CHECK_LAUNCH("dwconv");
  return LMV_OK;
}

// image rows per workgroup of the weight-gradient kernel: aim at <= ~1024 workgroups of >= 8 pixels per thread
inline int bwd_w_rows_per_block(int B, int H, int W, int slots) {
  int rpb = (8 * slots + W - 1) / W;
  if (rpb < 1) rpb = 1;
  while ((int64_t)B * ((H + rpb - 1) / rpb) > 1024 && rpb < H) ++rpb;
  return rpb;
}

}  // namespace

extern "C" int lmv_dwconv3x3_residual_fwd(const void* x, const float* weight, const float* bias, void* y, int B, int H, int W, int C,
                                          int dtype, void* stream) {
  if (int rc = check("dwconv_fwd", x, y, B, H, W, C, dtype)) return rc;
  if (!weight) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_fwd: null weight");
  return launch_dwconv<false>(x, weight, bias, y, B, H, W, C, dtype, (hipStream_t)stream);
}

extern "C" int lmv_dwconv3x3_residual_bwd_data(const void* dy, const float* weight, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
  if (int rc = check("dwconv_bwd_data", dy, dx, B, H, W, C, dtype)) return rc;
  if (!weight) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_data: null weight");
  return launch_dwconv<true>(dy, weight, nullptr, dx, B, H, W, C, dtype, (hipStream_t)stream);
}

extern "C" size_t lmv_dwconv3x3_bwd_weight_workspace_bytes(int B, int H, int W, int C, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  if (nch < 1 || nch > TPB) return 0;
  const int rpb = bwd_w_rows_per_block(B, H, W, TPB / nch);
  return (size_t)B * ((H + rpb - 1) / rpb) * 10 * C * sizeof(float);
}

extern "C" int lmv_dwconv3x3_bwd_weight(const void* dy, const void* x, float* dweight, float* dbias, int B, int H, int W, int C,
                                        void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (int rc = check("dwconv_bwd_weight", dy, x, B, H, W, C, dtype)) return rc;
  if (!dweight || !dbias) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_weight: null gradient buffer");
  hipStream_t st = (hipStream_t)stream;
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  if (nch > TPB) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_weight: C=%d too wide", C);
  const int slots = TPB / nch, threads = slots * nch;
  const int rpb = bwd_w_rows_per_block(B, H, W, slots);
  const int gx = (H + rpb - 1) / rpb, blocks = gx * B;
  const size_t need = (size_t)blocks * 10 * C * sizeof(float);
  if (!workspace || workspace_bytes < need) LMV_FAIL(LMV_ERR_WORKSPACE, "dwconv_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
  float* partial = reinterpret_cast<float*>(workspace);
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_bwd_w_kernel<bf16_t>), dim3(gx, B), dim3(threads), 0, st, (const bf16_t*)dy, (const bf16_t*)x, partial, H, W, C, slots, rpb);
  else hipLaunchKernelGGL((dwconv_bwd_w_kernel<float>), dim3(gx, B), dim3(threads), 0, st, (const float*)dy, (const float*)x, partial, H, W, C, slots, rpb);
  LMV_CHECK_LAUNCH("dwconv_bwd_weight");
  return lmv_launch_partial_reduce(partial, blocks, 10 * C, dweight, C, dbias, 1, st);
}
