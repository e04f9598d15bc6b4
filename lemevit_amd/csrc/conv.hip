// conv.hip -- conditional position embedding of a LeMeBlock: y = x + dwconv3x3(x) + bias on a
// token-major (NHWC) tensor (models/lemevit.py:510,546), plus its two backward kernels.
//
// Memory-bound.  A thread owns a RUN of R = 4 consecutive pixels of one image row x 16 bytes of channels and slides
// over the R + 2 input columns: every loaded 16-byte chunk feeds up to three outputs, so a pixel costs 4.5 loads
// instead of 9 (L1 / TA traffic is what limits this op; HBM traffic is one read + one write).
// Depth-wise weights are staged once per workgroup into LDS as [tap][C] fp32 and then held in registers.
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXC = 1024;
constexpr int R = 4;

struct RunIdx { int b, h, w0, c0; bool ok; };
template <int EPC>
__device__ __forceinline__ RunIdx decode_run(unsigned idx, int B, int H, int W, int C) {
  const unsigned nch = C / EPC, rw = (W + R - 1) / R;
  RunIdx r;
  const unsigned run = idx / nch;
  r.c0 = (idx - run * nch) * EPC;
  const unsigned t = run / rw;
  r.w0 = (run - t * rw) * R;
  r.b = t / H;
  r.h = t - r.b * H;
  r.ok = r.b < (unsigned)B;
  return r;
}

// FLIP = false: y = x + bias + sum_t w[c][t] * x[h+dy-1][w+dx-1]
// FLIP = true : y = x +        sum_t w[c][8-t] * x[h+dy-1][w+dx-1]   (transpose conv = backward-data)
template <typename T, bool FLIP>
__global__ __launch_bounds__(TPB) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                    T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float sw[10 * MAXC];                       // [9 taps | bias][C]
  for (int i = threadIdx.x; i < 9 * C; i += TPB) {
    const int c = i / 9, t = i - c * 9;
    sw[(FLIP ? 8 - t : t) * C + c] = wgt[i];
  }
  if (!FLIP && bias)
    for (int c = threadIdx.x; c < C; c += TPB) sw[9 * C + c] = bias[c];
  __syncthreads();
  const RunIdx ri = decode_run<EPC>(blockIdx.x * TPB + threadIdx.x, B, H, W, C);
  if (!ri.ok) return;
  // all 3 x (R + 2) input chunks are loaded UNCONDITIONALLY from clamped coordinates (so the loads issue back to
  // back instead of one L2 round trip per bounds branch); out-of-image taps are zeroed by a 0/1 factor.
  const int64_t img = (int64_t)ri.b * H * W;
  uint4 raw[3][R + 2];
  float rowm[3], colm[R + 2];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int hh = ri.h + ky - 1, hc = min(max(hh, 0), H - 1);
    rowm[ky] = (hh == hc) ? 1.f : 0.f;
#pragma unroll
    for (int cc = 0; cc < R + 2; ++cc) {
      const int wc = ri.w0 - 1 + cc, wcl = min(max(wc, 0), W - 1);
      if (ky == 0) colm[cc] = (wc == wcl) ? 1.f : 0.f;
      raw[ky][cc] = *reinterpret_cast<const uint4*>(x + (img + (int64_t)hc * W + wcl) * C + ri.c0);
    }
  }
  float acc[R][EPC];
#pragma unroll
  for (int p = 0; p < R; ++p)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[p][e] = (!FLIP && bias) ? sw[9 * C + ri.c0 + e] : 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int cc = 0; cc < R + 2; ++cc) {                 // input column w0 - 1 + cc
      float v[EPC];
      chunk_to_f<T>(raw[ky][cc], v);
      const float m = rowm[ky] * colm[cc];
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] *= m;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {                   // output pixel p = cc - kx uses tap (ky, kx)
        const int p = cc - kx;
        if (p < 0 || p >= R) continue;
        const float* wt = sw + (ky * 3 + kx) * C + ri.c0;
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[p][e] += wt[e] * v[e];
        if (ky == 1 && kx == 1) {                        // residual: + x itself
#pragma unroll
          for (int e = 0; e < EPC; ++e) acc[p][e] += v[e];
        }
      }
    }
  }
  const int64_t rowbase = img + (int64_t)ri.h * W;
#pragma unroll
  for (int p = 0; p < R; ++p)
    if (ri.w0 + p < W) *reinterpret_cast<uint4*>(y + (rowbase + ri.w0 + p) * C + ri.c0) = f_to_chunk<T>(acc[p]);
}

// dw[c][t] += sum_pix dy[pix][c] * x[pix + tap t][c] ; db[c] += sum_pix dy[pix][c]
// blockDim = nch * slots; thread (slot, chunk) walks pixel runs  slot, slot + stride, ... and keeps 10 x EPC running
// sums; the slots' sums are combined through LDS (one tap at a time) and each workgroup writes ONE partial row
// [10][C] -- no atomics (partial_reduce_kernel finishes).
template <typename T>
__global__ __launch_bounds__(TPB) void dwconv_bwd_w_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ partial,
                                                          int B, int H, int W, int C, int slots) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float sred[TPB * 8];
  const int nch = C / EPC;
  const int ch = threadIdx.x % nch, slot = threadIdx.x / nch, c0 = ch * EPC;
  float a[10][EPC];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[t][e] = 0.f;
  const unsigned rw = (W + R - 1) / R, nruns = (unsigned)B * H * rw;
  for (unsigned run = blockIdx.x * slots + slot; run < nruns; run += gridDim.x * slots) {
    const unsigned t = run / rw;
    const int w0 = (run - t * rw) * R, b = t / H, h = t - b * H;
    const int64_t img = (int64_t)b * H * W, rowbase = img + (int64_t)h * W;
    uint4 raw[3][R + 2], graw[R];
    float rowm[3], colm[R + 2];
#pragma unroll
    for (int p = 0; p < R; ++p) graw[p] = *reinterpret_cast<const uint4*>(dy + (rowbase + min(w0 + p, W - 1)) * C + c0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int hh = h + ky - 1, hc = min(max(hh, 0), H - 1);
      rowm[ky] = (hh == hc) ? 1.f : 0.f;
#pragma unroll
      for (int cc = 0; cc < R + 2; ++cc) {
        const int wc = w0 - 1 + cc, wcl = min(max(wc, 0), W - 1);
        if (ky == 0) colm[cc] = (wc == wcl) ? 1.f : 0.f;
        raw[ky][cc] = *reinterpret_cast<const uint4*>(x + (img + (int64_t)hc * W + wcl) * C + c0);
      }
    }
    float g[R][EPC];
#pragma unroll
    for (int p = 0; p < R; ++p) {
      chunk_to_f<T>(graw[p], g[p]);
      const float m = (w0 + p < W) ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < EPC; ++e) { g[p][e] *= m; a[9][e] += g[p][e]; }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int cc = 0; cc < R + 2; ++cc) {
        float v[EPC];
        chunk_to_f<T>(raw[ky][cc], v);
        const float m = rowm[ky] * colm[cc];
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] *= m;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int p = cc - kx;
          if (p < 0 || p >= R) continue;
#pragma unroll
          for (int e = 0; e < EPC; ++e) a[ky * 3 + kx][e] += g[p][e] * v[e];
        }
      }
    }
  }
  float* prow = partial + (int64_t)blockIdx.x * 10 * C;
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
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || C > MAXC || (int64_t)B * H * W * C >= ((int64_t)1 << 31))
    LMV_FAIL(LMV_ERR_SHAPE, "%s: bad shape B=%d H=%d W=%d C=%d (C %% 8 == 0, C <= %d, < 2^31 elements)", name, B, H, W, C, MAXC);
  if (!a || !b || !lmv_aligned16(a) || !lmv_aligned16(b)) LMV_FAIL(LMV_ERR_SHAPE, "%s: null or misaligned operand", name);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "%s: unsupported dtype %d", name, dtype);
  return LMV_OK;
}

template <bool FLIP>
int launch_dwconv(const void* x, const float* weight, const float* bias, void* y, int B, int H, int W, int C, int dtype, hipStream_t st) {
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  const int64_t threads = (int64_t)B * H * ((W + R - 1) / R) * nch;
  dim3 grid((unsigned)((threads + TPB - 1) / TPB)), block(TPB);
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_kernel<bf16_t, FLIP>), grid, block, 0, st, (const bf16_t*)x, weight, bias, (bf16_t*)y, B, H, W, C);
  else hipLaunchKernelGGL((dwconv_kernel<float, FLIP>), grid, block, 0, st, (const float*)x, weight, bias, (float*)y, B, H, W, C);
  LMV_CHECK_LAUNCH("dwconv");
  return LMV_OK;
}

// workgroups of the weight-gradient kernel: >= 4 runs per thread, at most 1024
inline int bwd_w_blocks(int B, int H, int W, int slots) {
  const int64_t nruns = (int64_t)B * H * ((W + R - 1) / R);
  int64_t blocks = (nruns + 4 * slots - 1) / (4 * slots);
  if (blocks > 1024) blocks = 1024;
  return (int)(blocks < 1 ? 1 : blocks);
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
  return (size_t)bwd_w_blocks(B, H, W, TPB / nch) * 10 * C * sizeof(float);
}

extern "C" int lmv_dwconv3x3_bwd_weight(const void* dy, const void* x, float* dweight, float* dbias, int B, int H, int W, int C,
                                        void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (int rc = check("dwconv_bwd_weight", dy, x, B, H, W, C, dtype)) return rc;
  if (!dweight || !dbias) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_weight: null gradient buffer");
  hipStream_t st = (hipStream_t)stream;
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  if (nch > TPB) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_weight: C=%d too wide", C);
  const int slots = TPB / nch, threads = slots * nch;
  const int blocks = bwd_w_blocks(B, H, W, slots);
  const size_t need = (size_t)blocks * 10 * C * sizeof(float);
  if (!workspace || workspace_bytes < need) LMV_FAIL(LMV_ERR_WORKSPACE, "dwconv_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
  float* partial = reinterpret_cast<float*>(workspace);
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_bwd_w_kernel<bf16_t>), dim3(blocks), dim3(threads), 0, st, (const bf16_t*)dy, (const bf16_t*)x, partial, B, H, W, C, slots);
  else hipLaunchKernelGGL((dwconv_bwd_w_kernel<float>), dim3(blocks), dim3(threads), 0, st, (const float*)dy, (const float*)x, partial, B, H, W, C, slots);
  LMV_CHECK_LAUNCH("dwconv_bwd_weight");
  return lmv_launch_partial_reduce(partial, blocks, 10 * C, dweight, C, dbias, 1, st);
}
