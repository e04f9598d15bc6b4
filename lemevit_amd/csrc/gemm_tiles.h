// gemm_tiles.h -- device building blocks shared by the GEMM kernels (gemm.hip) and the fused LayerNorm / MLP kernels (fused.hip):
// LDS panel images and their swizzles, LDS-DMA panel loads, MFMA fragment reads (C++ and inline-asm forms), one k-tile of MFMAs,
// and the coalesced per-wave epilogue.  See the header comment of gemm.hip for the design.
#pragma once
#include <utility>
#include "common.h"

namespace {

constexpr int PANEL = 128;   // rows (normal) or columns (transposed) of one LDS operand panel

template <int NWM_, int NWN_, int WM_>
struct Cfg {
  static constexpr int NWM = NWM_, NWN = NWN_, WM = WM_;      // waves along M / N, 16-row MFMA tiles per wave along M
  static constexpr int NW = NWM * NWN, NTHR = NW * 64;
  static constexpr int BM = NWM * WM * 16, BN = NWN * 64;
  static constexpr int PA = BM / PANEL, PB = BN / PANEL;      // operand panels per k-tile
};
using C128 = Cfg<2, 2, 4>;        // 128 x 128, 4 waves of 64 x 64
using C128w8 = Cfg<4, 2, 2>;      // 128 x 128, 8 waves of 32 x 64 (64-deep k-tiles at 4 waves per SIMD)

struct Problem {
  const void* a; const void* b; const float* bias; const void* res; const float* row_scale; const void* aux;
  void* out; void* out_pre; float* bias_grad;
  int M;           // output rows
  int Kred;        // reduction length
  int rps;         // rows per sample (DropPath scale index)
  int tiles_m;
  int tile_begin;  // first logical tile of this problem
  int pad_;
};
// one slab (or partial-row) reduction: out[i] += sum_s ws[s][i] (gemm.hip: splitk_reduce_kernel, lmv_reduce_batch, and the reduction that rides in a weight-gradient launch)
struct ReduceSegDev { const float* ws; float* out_w; float* out_b; int64_t stride, nw; int nslabs, nb, sl, blk0, kind, mode; };
// Implicit-GEMM form of Conv2d(Cin, Cout, 3, stride 2, padding 1) on an NHWC map (round 6; lmv_conv3x3s2_fwd / _dw): the "patch matrix" operand [B Ho Wo, KP] (column
// (ky * 3 + kx) * Cin + ci, zeros behind 9 Cin) is never materialised -- the LDS-DMA loads of a k-tile take their 16-byte pieces straight from the map (per-lane source address:
// the panel image in LDS is the one the plain GEMM builds), padding taps and the pad columns from a page of zeros.  Divisions by Cin / Wo / Ho Wo are multiplications by
// ceil(2^40 / d) (exact for n d < 2^40: n < 2^22 rows, d < 2^13).
struct ConvGeo {
  int on, H, W, Cin, Ho, Wo, HoWo, pad_;
  unsigned long long m_cin, m_wo, m_howo;
  const void* x;          // the map [B, H, W, Cin]
};
struct GemmArgs {
  Problem p[2];
  int nprob, N, lda, ldb, ldc, act, tiles_n, kt_per_split;
  ConvGeo cv;
  int cumap;                     // fwd / dX: CU-aware tile order (see gemm_kernel)
  int ntiles, nsplits, concat;   // dW: tiles of dW, k-splits, and whether problem 1's rows extend problem 0's reduction
  float ln_eps;           // LayerNorm-folded forward (lmv_ln_linear_fwd): eps of the folded LayerNorm
#ifdef LMV_GEMM_TIMING
  unsigned long long* dbg;      // s_memtime stamps of two workgroups (tools/gemm_timeline.py)
#endif
  float* ws;              // split-K (dW) mode: partial slabs [slab][N*K + N] fp32
  int64_t slab_stride;    // floats per slab
  int slab_base[2];       // first slab of each problem
};

// chunk swizzle of a reduction-contiguous panel whose rows are ROWB bytes (64 or 128)
template <int ROWB> __device__ __forceinline__ int swz_n(int row) { return ROWB == 64 ? ((-(row >> 2)) & 3) : ((row >> 1) & 7); }
__device__ __forceinline__ int swz_t(int r) { return (r & 3) | ((r >> 1) & 4); }

// ---- register-staged path (128x128 tile, 256 threads): global -> VGPR -> swizzled LDS -------------------------
// Normal operand: panel [128 rows][BK] (reduction-contiguous).  TR operand: panel [BK rows][128 cols].
template <typename T, bool TR, int BK, int NCH>
__device__ __forceinline__ void stage_load(uint4 (&regs)[NCH], const T* __restrict__ base, int64_t ld, int dim, int Kred,
                                           int tile0, int k0, int tid) {
  constexpr int EPC = DT<T>::EPC;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256;
    int64_t off; bool ok;
    if (!TR) {
      constexpr int CPR = BK / EPC;
      const int row = tile0 + c / CPR, k = k0 + (c % CPR) * EPC;
      ok = (row < dim) && (k < Kred);
      off = (int64_t)row * ld + k;
    } else {
      constexpr int CPR = PANEL / EPC;
      const int r = k0 + c / CPR, col = tile0 + (c % CPR) * EPC;
      ok = (r < Kred) && (col < dim);
      off = (int64_t)r * ld + col;
    }
    regs[i] = ok ? *reinterpret_cast<const uint4*>(base + off) : make_uint4(0, 0, 0, 0);
  }
}

template <typename T, bool TR, int BK, int NCH>
__device__ __forceinline__ void stage_store(unsigned char* s, const uint4 (&regs)[NCH], int tid) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int ROWB = BK * (int)sizeof(T);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256;
    int off;
    if (!TR) {
      constexpr int CPR = BK / EPC;
      const int row = c / CPR, kc = c % CPR;
      off = row * ROWB + ((kc ^ swz_n<ROWB>(row)) << 4);
    } else {
      constexpr int CPR = PANEL / EPC;
      const int r = c / CPR, cc = c % CPR;
      if (sizeof(T) == 2) off = r * 256 + ((((cc >> 1) ^ swz_t(r))) << 5) + ((cc & 1) << 4);
      else off = r * 512 + (cc << 4);
    }
    *reinterpret_cast<uint4*>(s + off) = regs[i];
  }
}

// ---- LDS-DMA path (bf16): one panel, NW waves ----------------------------------------------------------------
// Each wave-instruction fills one LINEAR 1 KiB segment of the panel image; the lane that lands on physical chunk p
// of row r fetches logical chunk p ^ swz(r).  Rows / columns outside the matrix are CLAMPED (never masked), so the
// duplicated data only reaches accumulators of outputs that are not stored; the reduction dimension is never
// ragged on this path (host guarantees Kred % BK == 0).
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

template <bool TR, int BK, int NW>
__device__ __forceinline__ void panel_dma(unsigned char* panel, const bf16_t* __restrict__ base, int64_t ld, int dim, int tile0, int k0,
                                          int lane, int wave) {
  constexpr int NSEG = PANEL * BK * 2 / 1024;
  static_assert(NSEG % NW == 0, "segments must divide over the waves");
#pragma unroll
  for (int i = 0; i < NSEG / NW; ++i) {
    const int sg = wave + i * NW;
    const bf16_t* src;
    if (!TR) {
      constexpr int ROWB = BK * 2, CPR = ROWB / 16, RPS = 1024 / ROWB;
      const int r = sg * RPS + lane / CPR, p = lane % CPR, kc = p ^ swz_n<ROWB>(r);
      src = base + (int64_t)min(tile0 + r, dim - 1) * ld + k0 + kc * 8;
    } else {
      const int r = sg * 4 + (lane >> 4), p16 = lane & 15, c16 = ((((p16 >> 1) ^ swz_t(r))) << 1) | (p16 & 1);
      src = base + (int64_t)(k0 + r) * ld + min(tile0 + c16 * 8, dim - 8);
    }
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(panel + sg * 1024), 16, 0, 0);
  }
}

// The same panel images, gathered from an NHWC map (ConvGeo): the operand element (row = output pixel, column k = tap * Cin + ci) is x[b, 2 oy + ky - 1, 2 ox + kx - 1, ci], zero
// outside the map and for k >= 9 Cin.  A 16-byte piece (8 channels) never straddles a tap because Cin % 8 == 0.  Not TR: the forward operand (rows = pixels, reduction over k);
// TR: the weight-gradient operand (reduction over pixels, columns = k).
__device__ __attribute__((aligned(128))) unsigned lmv_zero_page[32];
__device__ __forceinline__ unsigned cv_div(unsigned n, unsigned long long m) { return (unsigned)(((unsigned long long)n * m) >> 40); }
__device__ __forceinline__ const bf16_t* cv_src(const ConvGeo& cv, int pixel, int k) {
  const unsigned b = cv_div((unsigned)pixel, cv.m_howo), rem = (unsigned)pixel - b * (unsigned)cv.HoWo, oy = cv_div(rem, cv.m_wo), ox = rem - oy * (unsigned)cv.Wo;
  const unsigned tap = cv_div((unsigned)k, cv.m_cin), ci = (unsigned)k - tap * (unsigned)cv.Cin, ky = (tap * 11u) >> 5, kx = tap - 3u * ky;
  const int iy = 2 * (int)oy - 1 + (int)ky, ix = 2 * (int)ox - 1 + (int)kx;
  const bool ok = tap < 9u && (unsigned)iy < (unsigned)cv.H && (unsigned)ix < (unsigned)cv.W;
  const int64_t off = ((int64_t)((int)b * cv.H + iy) * cv.W + ix) * cv.Cin + (int)ci;
  return ok ? reinterpret_cast<const bf16_t*>(cv.x) + off : reinterpret_cast<const bf16_t*>(lmv_zero_page);
}
template <bool TR, int BK, int NW>
__device__ __forceinline__ void panel_dma_conv(unsigned char* panel, const ConvGeo& cv, int dim, int tile0, int k0, int lane, int wave) {
  constexpr int NSEG = PANEL * BK * 2 / 1024;
  static_assert(NSEG % NW == 0, "segments must divide over the waves");
#pragma unroll
  for (int i = 0; i < NSEG / NW; ++i) {
    const int sg = wave + i * NW;
    const bf16_t* src;
    if (!TR) {
      constexpr int ROWB = BK * 2, CPR = ROWB / 16, RPS = 1024 / ROWB;
      const int r = sg * RPS + lane / CPR, p = lane % CPR, kc = p ^ swz_n<ROWB>(r);
      src = cv_src(cv, min(tile0 + r, dim - 1), k0 + kc * 8);
    } else {
      const int r = sg * 4 + (lane >> 4), p16 = lane & 15, c16 = ((((p16 >> 1) ^ swz_t(r))) << 1) | (p16 & 1);
      src = cv_src(cv, k0 + r, min(tile0 + c16 * 8, dim - 8));
    }
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(panel + sg * 1024), 16, 0, 0);
  }
}

// ---- LDS -> MFMA fragments (row / column `base` inside ONE panel) ------------------------------------------------
template <bool TR, int BK>
__device__ __forceinline__ bf16x8_t frag_bf16(const unsigned char* s, int base, int lane, int hh) {
  if (!TR) {
    constexpr int ROWB = BK * 2;
    const int row = base + (lane & 15), kc = hh * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8_t*>(s + row * ROWB + ((kc ^ swz_n<ROWB>(row)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, rr = i >> 2, q = i & 3;
    const int r1 = hh * 32 + g * 8 + rr, r2 = r1 + 4, ch = base >> 4;
    typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;
    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(s + r1 * 256 + ((ch ^ swz_t(r1)) << 5) + q * 8));
    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(s + r2 * 256 + ((ch ^ swz_t(r2)) << 5) + q * 8));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
// fp32: k = hh*16 + (lane>>4)*4 + s for s = 0..3
template <bool TR>
__device__ __forceinline__ void frag_f32(const unsigned char* s, int base, int lane, int hh, float (&f)[4]) {
  if (!TR) {
    const int row = base + (lane & 15), kc = hh * 4 + (lane >> 4);
    const float4 v = *reinterpret_cast<const float4*>(s + row * 128 + ((kc ^ swz_n<128>(row)) << 4));
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    const int col = base + (lane & 15), r0 = hh * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const float*>(s + (r0 + q) * 512 + col * 4);
  }
}

template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- the same fragments through inline asm (LDS-DMA loop) -----------------------------------------------------------
// hipcc orders every LDS read it can SEE behind all pending LDS-DMA writes with s_waitcnt vmcnt(0) -- in a double-buffered loop that
// is the k-tile just requested, i.e. the prefetch never overlaps the MFMAs of the workgroup that issued it.  The DMA loop therefore
// reads its fragments with asm ds_read (byte offsets into LDS) and places its own counted waits.
template <bool TR, int BK>
__device__ __forceinline__ unsigned frag_off(int base, int lane, int hh) {          // byte offset of the lane's (first) read inside the panel
  if (!TR) {
    constexpr int ROWB = BK * 2;
    const int row = base + (lane & 15), kc = hh * 4 + (lane >> 4);
    return row * ROWB + ((kc ^ swz_n<ROWB>(row)) << 4);
  } else {
    const int g = lane >> 4, i = lane & 15, rr = i >> 2, q = i & 3;
    const int r1 = hh * 32 + g * 8 + rr, ch = base >> 4;
    return r1 * 256 + ((ch ^ swz_t(r1)) << 5) + q * 8;                              // second read: row r1 + 4 = + 1024 bytes, same swizzle
  }
}
template <bool TR>
__device__ __forceinline__ bf16x8_t frag_ld(unsigned addr) {
  if (!TR) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
  } else {
    bf16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(addr) : "memory");
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
template <int N_> __device__ __forceinline__ void wait_vm() {
  static_assert(N_ >= 0 && N_ < 64, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

#ifndef LMV_DUAL_HALF
#define LMV_DUAL_HALF 1
#endif
template <int N_> __device__ __forceinline__ void wait_lgkm() {
  static_assert(N_ >= 0 && N_ < 16, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory");
}

// Row statistics of the token operand for the LayerNorm-folded forward (fused.hip / lmv_ln_linear_fwd): every MFMA fragment of the token
// rows passes through here once, so sum and sum of squares of a row accumulate beside the MFMAs (v_dot2_f32_bf16: 8 VALU ops per fragment);
// lane -> row (lane & 15) of row tile t, the four lane groups hold the four k-quarters and are combined by finish().
struct NoStat { static constexpr bool on = false; };
template <int WM> struct RowStat {
  static constexpr bool on = true;
  float s[WM], q[WM];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int t = 0; t < WM; ++t) s[t] = q[t] = 0.f;
  }
  __device__ __forceinline__ void add(int t, const bf16x8_t& v) {
    const bf16x2_t one = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
    // (pairs taken with shufflevector: indexing a bit-cast <4 x i32> view of the fragment made hipcc feed element 0 to all four dots)
    const bf16x2_t p0 = __builtin_shufflevector(v, v, 0, 1), p1 = __builtin_shufflevector(v, v, 2, 3);
    const bf16x2_t p2 = __builtin_shufflevector(v, v, 4, 5), p3 = __builtin_shufflevector(v, v, 6, 7);
    s[t] = __builtin_amdgcn_fdot2_f32_bf16(p0, one, s[t], false); q[t] = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, q[t], false);
    s[t] = __builtin_amdgcn_fdot2_f32_bf16(p1, one, s[t], false); q[t] = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, q[t], false);
    s[t] = __builtin_amdgcn_fdot2_f32_bf16(p2, one, s[t], false); q[t] = __builtin_amdgcn_fdot2_f32_bf16(p2, p2, q[t], false);
    s[t] = __builtin_amdgcn_fdot2_f32_bf16(p3, one, s[t], false); q[t] = __builtin_amdgcn_fdot2_f32_bf16(p3, p3, q[t], false);
  }
  __device__ __forceinline__ void add(int t, float v) { s[t] += v; q[t] = fmaf(v, v, q[t]); }
  // -> s[t] = mean, q[t] = rstd of row (lane & 15) of row tile t (one-pass variance in fp32, clamped at 0)
  __device__ __forceinline__ void finish(int K, float eps) {
#pragma unroll
    for (int t = 0; t < WM; ++t) {
      float a = s[t], b = q[t];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      const float mean = a / K, var = fmaxf(b / K - mean * mean, 0.f);
      s[t] = mean; q[t] = rsqrtf(var + eps);
    }
  }
};

// MFMAs of one 32-deep half k-tile whose fragment reads are in flight, followed by PEND younger LDS reads (LDS returns in order).
// ONE wait for the whole half: waiting per A row (counted lgkmcnt before each group of 4 MFMAs) measured 1.5 % slower per train step.
template <bool BSUM, int WM, int PEND, typename RS = NoStat>
__device__ __forceinline__ void mma_half(bf16x8_t (&af)[WM], bf16x8_t (&bf)[4], f32x4_t (&acc)[WM][4], f32x4_t (&accb)[WM], bool do_bsum, RS* rs = nullptr) {
  wait_lgkm<PEND>();
#pragma unroll
  for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(bf[t]));          // the MFMAs below must not be scheduled above the wait
#pragma unroll
  for (int t = 0; t < WM; ++t) asm volatile("" : "+v"(af[t]));
#pragma unroll
  for (int ti = 0; ti < WM; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tj], af[ti], acc[ti][tj], 0, 0, 0);
  if constexpr (RS::on) {
#pragma unroll
    for (int ti = 0; ti < WM; ++ti) rs->add(ti, af[ti]);
  }
  if constexpr (BSUM) {
    if (do_bsum) {          // wave-uniform: column sums of the A tile = A^T * ones
      typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
      const u16x8_t o16 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
      const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, o16);
#pragma unroll
      for (int ti = 0; ti < WM; ++ti) accb[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[ti], accb[ti], 0, 0, 0);
    }
  }
}

// one k-tile of the DMA loop: aA / aB = LDS byte address of the wave's A / B panel, oa / ob = its first row / column inside that panel
template <bool ATR, bool BTR, int BK, bool BSUM, int WM, typename RS = NoStat>
__device__ __forceinline__ void tile_mma_dma(unsigned aA, int oa, unsigned aB, int ob, f32x4_t (&acc)[WM][4], f32x4_t (&accb)[WM], bool do_bsum, int lane, RS* rs = nullptr) {
  constexpr int IA = ATR ? 2 : 1, IB = BTR ? 2 : 1, NH = 4 * IB + WM * IA;      // LDS instructions per fragment / per half k-tile
  auto rd = [&](bf16x8_t (&af)[WM], bf16x8_t (&bf)[4], int hh) {
#pragma unroll
    for (int t = 0; t < 4; ++t) bf[t] = frag_ld<BTR>(aB + frag_off<BTR, BK>(ob + t * 16, lane, hh));
#pragma unroll
    for (int t = 0; t < WM; ++t) af[t] = frag_ld<ATR>(aA + frag_off<ATR, BK>(oa + t * 16, lane, hh));
  };
  bf16x8_t af0[WM], bf0[4];
  rd(af0, bf0, 0);
  if constexpr (BK == 32) {
    mma_half<BSUM, WM, 0, RS>(af0, bf0, acc, accb, do_bsum, rs);
  } else if constexpr (LMV_DUAL_HALF && NH <= 15) {      // 64-deep: the second half's reads fly under the first half's MFMAs
    bf16x8_t af1[WM], bf1[4];
    rd(af1, bf1, 1);
    mma_half<BSUM, WM, NH, RS>(af0, bf0, acc, accb, do_bsum, rs);
    mma_half<BSUM, WM, 0, RS>(af1, bf1, acc, accb, do_bsum, rs);
  } else {
    mma_half<BSUM, WM, 0, RS>(af0, bf0, acc, accb, do_bsum, rs);
    rd(af0, bf0, 1);
    mma_half<BSUM, WM, 0, RS>(af0, bf0, acc, accb, do_bsum, rs);
  }
}

// one k-tile: sA / sB = the wave's A / B panel, oa / ob = its first row / column inside that panel
template <typename T, bool ATR, bool BTR, int BK, bool BSUM, int WM, typename RS = NoStat>
__device__ __forceinline__ void tile_mma(const unsigned char* sA, int oa, const unsigned char* sB, int ob, f32x4_t (&acc)[WM][4],
                                         f32x4_t (&accb)[WM], bool do_bsum, int lane, RS* rs = nullptr) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int hh = 0; hh < BK / 32; ++hh) {
      bf16x8_t af[WM], bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bf[t] = frag_bf16<BTR, BK>(sB, ob + t * 16, lane, hh);
#pragma unroll
      for (int t = 0; t < WM; ++t) af[t] = frag_bf16<ATR, BK>(sA, oa + t * 16, lane, hh);
#pragma unroll
      for (int ti = 0; ti < WM; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tj], af[ti], acc[ti][tj], 0, 0, 0);
      if constexpr (RS::on) {
#pragma unroll
        for (int ti = 0; ti < WM; ++ti) rs->add(ti, af[ti]);
      }
      if constexpr (BSUM) {
        if (do_bsum) {          // wave-uniform: column sums of the A tile = A^T * ones
          typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
          const u16x8_t o16 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
          const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, o16);
#pragma unroll
          for (int ti = 0; ti < WM; ++ti) accb[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[ti], accb[ti], 0, 0, 0);
        }
      }
    }
  } else {
    static_assert(sizeof(T) == 2 || BK == 32, "fp32 tiles are 32 deep");
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float af[WM][4], bf[4][4];
#pragma unroll
      for (int t = 0; t < WM; ++t) frag_f32<ATR>(sA, oa + t * 16, lane, hh, af[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) frag_f32<BTR>(sB, ob + t * 16, lane, hh, bf[t]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int ti = 0; ti < WM; ++ti)
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tj][q], af[ti][q], acc[ti][tj], 0, 0, 0);
        if constexpr (RS::on) {
#pragma unroll
          for (int ti = 0; ti < WM; ++ti) rs->add(ti, af[ti][q]);
        }
        if constexpr (BSUM) {
          if (do_bsum) {
#pragma unroll
            for (int ti = 0; ti < WM; ++ti) accb[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, af[ti][q], accb[ti], 0, 0, 0);
          }
        }
      }
    }
  }
}

template <typename T> __device__ __forceinline__ float act_gelu(float x);
template <> __device__ __forceinline__ float act_gelu<float>(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float act_gelu<bf16_t>(float x) { return gelu_fast_f(x); }
template <typename T> __device__ __forceinline__ float act_gelu_grad(float x);
template <> __device__ __forceinline__ float act_gelu_grad<float>(float x) { return gelu_grad_f(x); }
template <> __device__ __forceinline__ float act_gelu_grad<bf16_t>(float x) { return gelu_grad_fast_f(x); }

// ---- coalesced epilogue: fp32 tile -> swizzled LDS (64 rows per pass) -> 16-byte row-major stores --------------
template <typename T, typename CF, int REGION_BYTES> struct Epi {
  static constexpr int EPC = DT<T>::EPC;
  static constexpr int CPR = 64 / EPC;                 // output chunks per 64-column row
  static constexpr int RPP = REGION_BYTES >= 16384 ? 64 : (REGION_BYTES >= 8192 ? 32 : 16);
  static_assert(REGION_BYTES >= RPP * 256 && (CF::WM * 16) % RPP == 0, "per-wave epilogue region too small");
  static constexpr int NPASS = CF::WM * 16 / RPP, NITER = RPP * CPR / 64;
  // Everything the epilogue READS from global memory, issued for the wave's whole strip before the first LDS transpose: one
  // 16-byte chunk per output chunk (the GELU' operand u, or the residual), the DropPath scale of each chunk's row and the bias.
  // Issued at the point of use these loads stalled every pass for a memory round trip: 0.75 ms of a 33.6 ms Base train step.
  static constexpr bool HOIST_BIAS = CF::NW >= 8 && CF::WM == 2;     // bias + row scales: 24 more live registers -- the 4-wave kernels (128-register budget) would spill
  uint4 pf[NPASS][NITER];
  float rs[HOIST_BIAS ? NPASS : 1][HOIST_BIAS ? NITER : 1];
  float4 b4[HOIST_BIAS ? 4 : 1];

  __device__ __forceinline__ void prefetch(const Problem& P, int act, int N, int64_t ldc, int m0, int n0, int wm, int wn, int lane) {
    const int nw0 = n0 + wn * 64;
    const T* src = reinterpret_cast<const T*>(act == LMV_ACT_GELU_GRAD ? P.aux : P.res);      // wave-uniform
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
      for (int i = 0; i < NITER; ++i) {
        const int c = lane + i * 64, r = c / CPR, oc = c % CPR;
        const int m = min(m0 + wm * (CF::WM * 16) + p * RPP + r, P.M - 1), n = min(nw0 + oc * EPC, N - EPC);      // clamped, never masked
        if (src) pf[p][i] = *reinterpret_cast<const uint4*>(src + (int64_t)m * ldc + n);
        if constexpr (HOIST_BIAS) rs[p][i] = 1.0f;
      }
    if constexpr (HOIST_BIAS) {
      if (P.row_scale) {      // workgroup-uniform branch: the per-chunk integer division by rows-per-sample only where DropPath is live
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
          for (int i = 0; i < NITER; ++i) {
            const int c = lane + i * 64, r = c / CPR;
            const int m = min(m0 + wm * (CF::WM * 16) + p * RPP + r, P.M - 1);
            rs[p][i] = P.row_scale[m / P.rps];
          }
      }
    }
    if constexpr (HOIST_BIAS) {
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) b4[tj] = bias_at(P, N, nw0 + tj * 16 + (lane >> 4) * 4);
    }
  }
  static __device__ __forceinline__ float4 bias_at(const Problem& P, int N, int n) {
    return (P.bias && n < N) ? *reinterpret_cast<const float4*>(P.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // out_pre = acc + bias (optional pre-activation copy for the backward pass) and out = res + row_scale * act(acc + bias)
  // (act = GELU, or x GELU'(aux) for dX) leave in ONE pass: the transposed fp32 tile is read back once and stored twice.
  // Every wave transposes ITS OWN 64-column strip through a private LDS region (no workgroup barrier, all waves busy):
  // RPP rows per pass, a row of 64 fp32 = 16 float4 chunks, chunk c of row r stored at c ^ (r & 7).
  // LN = true (lmv_ln_linear_fwd): the accumulators hold x W'^T of the RAW rows; the folded LayerNorm is applied on the way into the
  // transpose: v = rstd[row] * (acc - mean[row] * colsum[n]) + bias'[n]  (ln_mean / ln_rstd: per row tile, row = lane & 15; colsum = P.aux)
  struct NoStamp { __device__ __forceinline__ void operator()() const {} };
  template <bool LN = false, typename ST = NoStamp>
  __device__ __forceinline__ void store(unsigned char* smem, const f32x4_t (&acc)[CF::WM][4], const Problem& P, int act, int N, int64_t ldc,
                                        int m0, int n0, int wm, int wn, int lane, int wave, const float* ln_mean = nullptr, const float* ln_rstd = nullptr,
                                        ST stamp = ST{}) const {
    float* sT = reinterpret_cast<float*>(smem + wave * REGION_BYTES);
    T* outp = reinterpret_cast<T*>(P.out);
    T* prep = reinterpret_cast<T*>(P.out_pre);
    const int nw0 = n0 + wn * 64;                 // first column of this wave's strip
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        float4 b;
        if constexpr (HOIST_BIAS) b = b4[tj]; else b = bias_at(P, N, nw0 + tj * 16 + (lane >> 4) * 4);
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LN) {
          const int n = nw0 + tj * 16 + (lane >> 4) * 4;
          if (n < N) cs = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(P.aux) + n);
        }
#pragma unroll
        for (int i = 0; i < RPP / 16; ++i) {
          f32x4_t a = acc[p * (RPP / 16) + i][tj];
          if constexpr (LN) {
            const float mu = ln_mean[p * (RPP / 16) + i], rs = ln_rstd[p * (RPP / 16) + i];
            a[0] = rs * (a[0] - mu * cs.x); a[1] = rs * (a[1] - mu * cs.y); a[2] = rs * (a[2] - mu * cs.z); a[3] = rs * (a[3] - mu * cs.w);
          }
          const float4 v = make_float4(a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
          const int r = i * 16 + (lane & 15), c4 = tj * 4 + (lane >> 4);
          *reinterpret_cast<float4*>(sT + r * 64 + ((c4 ^ (r & 7)) << 2)) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS writes are visible to all of its lanes
      __builtin_amdgcn_wave_barrier();
      stamp();
#pragma unroll
      for (int i = 0; i < NITER; ++i) {
        const int c = lane + i * 64, r = c / CPR, oc = c % CPR;
        const int m = m0 + wm * (CF::WM * 16) + p * RPP + r, n = nw0 + oc * EPC;
        if (m >= P.M || n >= N) continue;
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const int c4 = (oc * EPC + e) >> 2;
          const float4 t = *reinterpret_cast<const float4*>(sT + r * 64 + ((c4 ^ (r & 7)) << 2));
          v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
        }
        const int64_t o = (int64_t)m * ldc + n;
        if (prep) *reinterpret_cast<uint4*>(prep + o) = f_to_chunk<T>(v);      // pre-activation copy kept for the backward pass (same pass)
        if (act == LMV_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = act_gelu<T>(v[e]);
        } else if (act == LMV_ACT_GELU_GRAD) {
          float u[EPC];
          chunk_to_f<T>(pf[p][i], u);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] *= act_gelu_grad<T>(u[e]);
        }
        if (P.row_scale) {
          float sc;
          if constexpr (HOIST_BIAS) sc = rs[p][i]; else sc = P.row_scale[m / P.rps];
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] *= sc;
        }
        if (P.res) {
          float r8[EPC];
          if (act == LMV_ACT_GELU_GRAD) chunk_to_f<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(P.res) + o), r8);
          else chunk_to_f<T>(pf[p][i], r8);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] += r8[e];
        }
        *reinterpret_cast<uint4*>(outp + o) = f_to_chunk<T>(v);
        stamp();
      }
    }
  }
};

}  // namespace
