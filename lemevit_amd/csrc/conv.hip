// conv.hip -- conditional position embedding of a LeMeBlock: y = x + dwconv3x3(x) + bias on a
// token-major (NHWC) tensor (models/lemevit.py:510,546), plus its two backward kernels.
//
// Memory-bound (one read + one write of the map; the weight gradient reads two maps).  A thread owns 16 bytes of
// channels of a STRIP of R pixel columns (2; 1 in the weight gradient) and slides down V rows of it: the 3 x (R + 2) window of 16-byte chunks
// lives in registers, one new window row (R + 2 chunks) is fetched per output row -- issued a row ahead so it flies
// under the current row's FMAs -- and every chunk feeds up to nine outputs.  The 9 x 8 depth-wise weights of the
// thread's channels are 288 contiguous bytes of the [C][3][3] parameter: they go straight into registers (no LDS
// staging, no barrier; consecutive lanes own consecutive channel chunks, so map accesses are fully coalesced).
// V is picked so that the launch has >= ~2k wavefronts (tools/bench_kernels.py conv; LMV_DWCONV_V overrides).
// Out-of-image taps: coordinates are CLAMPED and the chunk is zeroed (integer AND) when its row enters the window, so
// the loads of a row issue back to back (a bounds branch around each load costs one L2 round trip per load).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXC = 2048;
constexpr int RF = 2;     // pixel columns per thread: forward / backward-data
constexpr int RW = 2;     // weight gradient: 10 x 8 running sums + a 3 x 4 window = 208 registers (2 waves / SIMD), but 3 loads per pixel
                          // instead of 4: 57 -> 43 us on the stage-1 maps (R = 1: 136 registers, 3 waves / SIMD)

struct Geo { int B, H, W, C, V, nranges, nstrips; };
struct Item { int b, h0, h1, w0; };

template <int R>
__device__ __forceinline__ Item decode_item(unsigned it, const Geo& g) {      // it = (b * nranges + range) * nstrips + strip
  const unsigned t = it / g.nstrips, strip = it - t * g.nstrips;
  Item r;
  r.b = t / g.nranges;
  r.h0 = (t - r.b * g.nranges) * g.V;
  r.h1 = min(g.H, r.h0 + g.V);
  r.w0 = strip * R;
  return r;
}

// window row h (clamped into the image) of strip w0: columns w0 - 1 .. w0 + R (clamped)
template <int R, typename T>
__device__ __forceinline__ void load_row(uint4 (&dst)[R + 2], const T* __restrict__ x, int64_t img, int h, const Geo& g, int w0, int c0) {
  const int hc = min(max(h, 0), g.H - 1);
  const T* rowp = x + (img + (int64_t)hc * g.W) * g.C + c0;
#pragma unroll
  for (int cc = 0; cc < R + 2; ++cc) {
    const int wcl = min(max(w0 - 1 + cc, 0), g.W - 1);
    dst[cc] = *reinterpret_cast<const uint4*>(rowp + (int64_t)wcl * g.C);
  }
}

// out-of-image window entries are ZEROED once, when a row enters the window (4 integer ANDs per chunk)
template <int R>
__device__ __forceinline__ void mask_row(uint4 (&row)[R + 2], int h, const Geo& g, const unsigned (&cmask)[R + 2]) {
  const unsigned rm = (h >= 0 && h < g.H) ? ~0u : 0u;
#pragma unroll
  for (int cc = 0; cc < R + 2; ++cc) {
    const unsigned m = rm & cmask[cc];
    row[cc].x &= m; row[cc].y &= m; row[cc].z &= m; row[cc].w &= m;
  }
}

// FLIP = false: y = x + bias + sum_t w[c][t] * x[h+dy-1][w+dx-1]
// FLIP = true : y = x +        sum_t w[c][8-t] * x[h+dy-1][w+dx-1]   (transpose conv = backward-data)
template <typename T, bool FLIP>
__global__ __launch_bounds__(TPB) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                    T* __restrict__ y, const Geo g) {
  constexpr int EPC = DT<T>::EPC, R = RF;
  const unsigned gid = blockIdx.x * TPB + threadIdx.x, nch = g.C / EPC;
  const unsigned it = gid / nch;
  const int c0 = (gid - it * nch) * EPC;
  if (it >= (unsigned)(g.B * g.nranges * g.nstrips)) return;
  const Item I = decode_item<R>(it, g);

  float wv[9 * EPC], bv[EPC];                             // wv[e * 9 + t] = w[c0 + e][t]
#pragma unroll
  for (int i = 0; i < 9 * EPC / 4; ++i) {
    const float4 t = *reinterpret_cast<const float4*>(wgt + (int64_t)c0 * 9 + i * 4);
    wv[i * 4] = t.x; wv[i * 4 + 1] = t.y; wv[i * 4 + 2] = t.z; wv[i * 4 + 3] = t.w;
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) { bv[e] = (!FLIP && bias) ? bias[c0 + e] : 0.f; wv[e * 9 + 4] += 1.f; }      // centre tap + 1: the residual
  unsigned cmask[R + 2];
#pragma unroll
  for (int cc = 0; cc < R + 2; ++cc) { const int wc = I.w0 - 1 + cc; cmask[cc] = (wc >= 0 && wc < g.W) ? ~0u : 0u; }

  const int64_t img = (int64_t)I.b * g.H * g.W;
  uint4 win[3][R + 2];
  load_row<R>(win[0], x, img, I.h0 - 1, g, I.w0, c0);
  load_row<R>(win[1], x, img, I.h0, g, I.w0, c0);
  load_row<R>(win[2], x, img, I.h0 + 1, g, I.w0, c0);
  mask_row<R>(win[0], I.h0 - 1, g, cmask);
  mask_row<R>(win[1], I.h0, g, cmask);
  mask_row<R>(win[2], I.h0 + 1, g, cmask);
  for (int h = I.h0; h < I.h1; ++h) {
    uint4 nxt[R + 2];
    if (h + 1 < I.h1) load_row<R>(nxt, x, img, h + 2, g, I.w0, c0);      // next window row: in flight during this row's math
    float acc[R][EPC];
#pragma unroll
    for (int p = 0; p < R; ++p)
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[p][e] = bv[e];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int cc = 0; cc < R + 2; ++cc) {                 // input column w0 - 1 + cc
        float v[EPC];
        chunk_to_f<T>(win[ky][cc], v);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {                   // output pixel p = cc - kx uses tap (ky, kx)
          const int p = cc - kx;
          if (p < 0 || p >= R) continue;
          const int t = FLIP ? 8 - (ky * 3 + kx) : ky * 3 + kx;
#pragma unroll
          for (int e = 0; e < EPC; ++e) acc[p][e] += wv[e * 9 + t] * v[e];
        }
      }
    }
    T* yrow = y + (img + (int64_t)h * g.W + I.w0) * g.C + c0;
#pragma unroll
    for (int p = 0; p < R; ++p)
      if (I.w0 + p < g.W) *reinterpret_cast<uint4*>(yrow + (int64_t)p * g.C) = f_to_chunk<T>(acc[p]);
    mask_row<R>(nxt, h + 2, g, cmask);
#pragma unroll
    for (int cc = 0; cc < R + 2; ++cc) { win[0][cc] = win[1][cc]; win[1][cc] = win[2][cc]; win[2][cc] = nxt[cc]; }
  }
}

// dw[c][t] += sum_pix dy[pix][c] * x[pix + tap t][c] ; db[c] += sum_pix dy[pix][c]
// blockDim = nch * slots; thread (slot, chunk) walks items  slot, slot + stride, ... (an item = strip x row range, slid
// over exactly as above with the x window and one dy row) and keeps 10 x EPC running sums; the slots' sums are combined
// through LDS (one tap at a time) and each workgroup writes ONE partial row [10][C] -- no atomics
// (partial_reduce_kernel finishes).
template <typename T>
__global__ __launch_bounds__(TPB) void dwconv_bwd_w_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ partial,
                                                          const Geo g, int slots) {
  constexpr int EPC = DT<T>::EPC, R = RW;
  __shared__ float sred[TPB * 8];
  const int nch = g.C / EPC;
  const int ch = threadIdx.x % nch, slot = threadIdx.x / nch, c0 = ch * EPC;
  float a[10][EPC];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[t][e] = 0.f;
  const unsigned nitems = (unsigned)(g.B * g.nranges * g.nstrips);
  for (unsigned it = blockIdx.x * slots + slot; it < nitems; it += gridDim.x * slots) {
    const Item I = decode_item<R>(it, g);
    unsigned cmask[R + 2], pixm[R];
#pragma unroll
    for (int cc = 0; cc < R + 2; ++cc) { const int wc = I.w0 - 1 + cc; cmask[cc] = (wc >= 0 && wc < g.W) ? ~0u : 0u; }
#pragma unroll
    for (int p = 0; p < R; ++p) pixm[p] = (I.w0 + p < g.W) ? ~0u : 0u;
    const int64_t img = (int64_t)I.b * g.H * g.W;
    auto load_dy = [&](uint4 (&dst)[R], int h) {
      const T* rowp = dy + (img + (int64_t)h * g.W) * g.C + c0;
#pragma unroll
      for (int p = 0; p < R; ++p) dst[p] = *reinterpret_cast<const uint4*>(rowp + (int64_t)min(I.w0 + p, g.W - 1) * g.C);
    };
    auto mask_dy = [&](uint4 (&r)[R]) {
#pragma unroll
      for (int p = 0; p < R; ++p) { r[p].x &= pixm[p]; r[p].y &= pixm[p]; r[p].z &= pixm[p]; r[p].w &= pixm[p]; }
    };
    uint4 win[3][R + 2], graw[R];
    load_row<R>(win[0], x, img, I.h0 - 1, g, I.w0, c0);
    load_row<R>(win[1], x, img, I.h0, g, I.w0, c0);
    load_row<R>(win[2], x, img, I.h0 + 1, g, I.w0, c0);
    load_dy(graw, I.h0);
    mask_row<R>(win[0], I.h0 - 1, g, cmask);
    mask_row<R>(win[1], I.h0, g, cmask);
    mask_row<R>(win[2], I.h0 + 1, g, cmask);
    mask_dy(graw);
    for (int h = I.h0; h < I.h1; ++h) {
      uint4 nxt[R + 2], gnxt[R];
      if (h + 1 < I.h1) { load_row<R>(nxt, x, img, h + 2, g, I.w0, c0); load_dy(gnxt, h + 1); }
      float gf[R][EPC];
#pragma unroll
      for (int p = 0; p < R; ++p) {
        chunk_to_f<T>(graw[p], gf[p]);
#pragma unroll
        for (int e = 0; e < EPC; ++e) a[9][e] += gf[p][e];
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int cc = 0; cc < R + 2; ++cc) {
          float v[EPC];
          chunk_to_f<T>(win[ky][cc], v);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int p = cc - kx;
            if (p < 0 || p >= R) continue;
#pragma unroll
            for (int e = 0; e < EPC; ++e) a[ky * 3 + kx][e] += gf[p][e] * v[e];
          }
        }
      }
      mask_row<R>(nxt, h + 2, g, cmask);
      mask_dy(gnxt);
#pragma unroll
      for (int cc = 0; cc < R + 2; ++cc) { win[0][cc] = win[1][cc]; win[1][cc] = win[2][cc]; win[2][cc] = nxt[cc]; }
#pragma unroll
      for (int p = 0; p < R; ++p) graw[p] = gnxt[p];
    }
  }
  float* prow = partial + (int64_t)blockIdx.x * 10 * g.C;
#pragma unroll                                  // (a[t] must stay a compile-time register index)
  for (int t = 0; t < 10; ++t) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPC; ++e) sred[slot * g.C + c0 + e] = a[t][e];       // [slots][C] (slots * C <= TPB * 8)
    __syncthreads();
    for (int c = threadIdx.x; c < g.C; c += blockDim.x) {
      float s = 0.f;
      for (int k = 0; k < slots; ++k) s += sred[k * g.C + c];
      prow[t * g.C + c] = s;
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

// rows per thread: as many as still leave ~2k wavefronts in the launch
inline Geo make_geo(int B, int H, int W, int C, int dtype, int R, bool wgrad = false) {
  Geo g;
  g.B = B; g.H = H; g.W = W; g.C = C;
  g.nstrips = (W + R - 1) / R;
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  if (lmv_config().dwconv_v > 0) {
    g.V = lmv_config().dwconv_v;
  } else if (wgrad) {
    g.V = H < 14 ? H : 14;      // weight gradient: long strips (more accumulation per thread, fewer partial rows) win on every stage
  } else {
    const int64_t per_range = (int64_t)B * g.nstrips * nch, want = 2048 * 64;
    int64_t nr = (want + per_range - 1) / per_range;
    if (nr < 1) nr = 1;
    g.V = (int)((H + nr - 1) / nr);
  }
  if (g.V < 1) g.V = 1;
  if (g.V > H) g.V = H;
  g.nranges = (H + g.V - 1) / g.V;
  return g;
}

template <bool FLIP>
int launch_dwconv(const void* x, const float* weight, const float* bias, void* y, int B, int H, int W, int C, int dtype, hipStream_t st) {
  const Geo g = make_geo(B, H, W, C, dtype, RF);
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  const int64_t threads = (int64_t)B * g.nranges * g.nstrips * nch;
  dim3 grid((unsigned)((threads + TPB - 1) / TPB)), block(TPB);
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_kernel<bf16_t, FLIP>), grid, block, 0, st, (const bf16_t*)x, weight, bias, (bf16_t*)y, g);
  else hipLaunchKernelGGL((dwconv_kernel<float, FLIP>), grid, block, 0, st, (const float*)x, weight, bias, (float*)y, g);
  LMV_CHECK_LAUNCH("dwconv");
  return LMV_OK;
}

// workgroups of the weight-gradient kernel: one item per thread slot, at most 1024 (= partial rows to reduce)
inline int bwd_w_blocks(const Geo& g, int slots) {
  const int64_t nitems = (int64_t)g.B * g.nranges * g.nstrips;
  int64_t blocks = (nitems + slots - 1) / slots;
  if (blocks > 1024) blocks = 1024;
  return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int lmv_dwconv3x3_residual_fwd(const void* x, const float* weight, const float* bias, void* y, int B, int H, int W, int C,
                                          int dtype, void* stream) {
  if (int rc = check("dwconv_fwd", x, y, B, H, W, C, dtype)) return rc;
  if (!weight || !lmv_aligned16(weight)) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_fwd: null or misaligned weight");
  return launch_dwconv<false>(x, weight, bias, y, B, H, W, C, dtype, (hipStream_t)stream);
}

extern "C" int lmv_dwconv3x3_residual_bwd_data(const void* dy, const float* weight, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
  if (int rc = check("dwconv_bwd_data", dy, dx, B, H, W, C, dtype)) return rc;
  if (!weight || !lmv_aligned16(weight)) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_data: null or misaligned weight");
  return launch_dwconv<true>(dy, weight, nullptr, dx, B, H, W, C, dtype, (hipStream_t)stream);
}

extern "C" size_t lmv_dwconv3x3_bwd_weight_workspace_bytes(int B, int H, int W, int C, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  if (nch < 1 || nch > TPB) return 0;
  return (size_t)bwd_w_blocks(make_geo(B, H, W, C, dtype, RW, true), TPB / nch) * 10 * C * sizeof(float);
}

namespace {
int dwconv_bwd_weight_launch(const char* who, const void* dy, const void* x, int B, int H, int W, int C, void* workspace, size_t workspace_bytes, int dtype,
                             hipStream_t st, int* rows) {
  if (int rc = check(who, dy, x, B, H, W, C, dtype)) return rc;
  const int nch = C / (dtype == LMV_BF16 ? 8 : 4);
  if (nch > TPB) LMV_FAIL(LMV_ERR_SHAPE, "%s: C=%d too wide", who, C);
  const Geo g = make_geo(B, H, W, C, dtype, RW, true);
  const int slots = TPB / nch, threads = slots * nch;
  const int blocks = bwd_w_blocks(g, slots);
  const size_t need = (size_t)blocks * 10 * C * sizeof(float);
  if (!workspace || workspace_bytes < need) LMV_FAIL(LMV_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, need);
  float* partial = reinterpret_cast<float*>(workspace);
  if (dtype == LMV_BF16) hipLaunchKernelGGL((dwconv_bwd_w_kernel<bf16_t>), dim3(blocks), dim3(threads), 0, st, (const bf16_t*)dy, (const bf16_t*)x, partial, g, slots);
  else hipLaunchKernelGGL((dwconv_bwd_w_kernel<float>), dim3(blocks), dim3(threads), 0, st, (const float*)dy, (const float*)x, partial, g, slots);
  LMV_CHECK_LAUNCH(who);
  *rows = blocks;
  return LMV_OK;
}
}  // namespace

extern "C" int lmv_dwconv3x3_bwd_weight(const void* dy, const void* x, float* dweight, float* dbias, int B, int H, int W, int C,
                                        void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (!dweight || !dbias) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_weight: null gradient buffer");
  int rows = 0;
  if (int rc = dwconv_bwd_weight_launch("dwconv_bwd_weight", dy, x, B, H, W, C, workspace, workspace_bytes, dtype, (hipStream_t)stream, &rows)) return rc;
  return lmv_launch_partial_reduce(reinterpret_cast<const float*>(workspace), rows, 10 * C, dweight, C, dbias, 1, (hipStream_t)stream);
}
extern "C" int lmv_dwconv3x3_bwd_weight_partial(const void* dy, const void* x, int B, int H, int W, int C, void* workspace, size_t workspace_bytes,
                                                int* partial_rows, int dtype, void* stream) {
  if (!partial_rows) LMV_FAIL(LMV_ERR_SHAPE, "dwconv_bwd_weight_partial: partial_rows must not be NULL");
  return dwconv_bwd_weight_launch("dwconv_bwd_weight_partial", dy, x, B, H, W, C, workspace, workspace_bytes, dtype, (hipStream_t)stream, partial_rows);
}
