// gemm.hip -- MFMA GEMM for the Linear layers of the LeMeViT blocks (gfx950).
//
// One kernel template covers the three contractions a Linear layer needs:
//   fwd  C[r,n] = sum_k  A[r,k]  W[n,k]      (A, W both reduction-contiguous: plain 16-B LDS reads)
//   dx   C[r,k] = sum_n dY[r,n]  W[n,k]      (W is reduction-MAJOR: ds_read_b64_tr_b16 transpose reads)
//   dw   C[n,k] = sum_r dY[r,n]  X[r,k]      (both operands reduction-major; the token reduction is split over
//                                              workgroups that write fp32 partial slabs, summed by a 2nd kernel:
//                                              deterministic, and no same-address atomics, which serialise at
//                                              ~170 ns each across XCDs)
// so the backward pass needs neither transposed weight copies nor transposed activations.
//
// Tile: 128 x 128 x BK per 256-thread workgroup (BK = 64 bf16 / 32 fp32 and K-tails), 4 waves as
// 2 x 2, each wave 64 x 64 = 4 x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16, or v_mfma_f32_16x16x4_f32 in
// the exact-fp32 mode).  Operands are register-staged (global -> VGPR -> LDS) into a DOUBLE-BUFFERED
// LDS image: the next k-tile's 16-B global loads are issued before the current tile's MFMAs and
// written to the other buffer after them, so there is ONE barrier per k-step.  LDS images are
// XOR-swizzled so that both the 16-B fragment reads and the transpose reads are bank-conflict free
// (derivation in DESIGN.md).  Workgroup ids are remapped so that the n-tiles of one m-tile (which
// re-read the same activation rows) run on the same XCD and hit its L2.
// The MFMA operands are passed swapped (W as "A") so every lane ends up with 4 CONSECUTIVE output
// columns of one row.  The epilogue stages the fp32 tile through (swizzled) LDS, 64 rows at a time, and
// leaves as whole 256-byte row segments: 16-byte coalesced stores, 16-byte coalesced residual / GELU'
// operand loads, with bias, exact-erf GELU, DropPath scale and residual add fused.
// In dW mode the bias gradient (column sums of dY) rides the matrix pipe: one extra MFMA per fragment
// against an all-ones operand.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NTHR = 256;

struct Problem {
  const void* a; const void* b; const float* bias; const void* res; const float* row_scale; const void* aux;
  void* out; void* out_pre; float* bias_grad;
  int M;           // output rows
  int Kred;        // reduction length
  int rps;         // rows per sample (DropPath scale index)
  int tiles_m;
  int tile_begin;  // first blockIdx.x of this problem
  int pad_;
};
struct GemmArgs {
  Problem p[2];
  int nprob, N, lda, ldb, ldc, act, tiles_n, kt_per_split;
  float* ws;              // split-K (dW) mode: partial slabs [slab][N*K + N] fp32
  int64_t slab_stride;    // floats per slab
  int slab_base[2];       // first slab of each problem
};

// chunk swizzle of a reduction-contiguous tile whose rows are ROWB bytes (64 or 128)
template <int ROWB> __device__ __forceinline__ int swz_n(int row) { return ROWB == 64 ? ((-(row >> 2)) & 3) : ((row >> 1) & 7); }
__device__ __forceinline__ int swz_t(int r) { return (r & 3) | ((r >> 1) & 4); }

// ---- global -> registers ---------------------------------------------------------------------
// Normal operand: tile [128 rows][BK] (reduction-contiguous).  TR operand: tile [BK rows][128 cols].
template <typename T, bool TR, int BK, int NCH>
__device__ __forceinline__ void stage_load(uint4 (&regs)[NCH], const T* __restrict__ base, int64_t ld, int dim, int Kred,
                                           int tile0, int k0, int tid) {
  constexpr int EPC = DT<T>::EPC;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTHR;
    int64_t off; bool ok;
    if (!TR) {
      constexpr int CPR = BK / EPC;
      const int row = tile0 + c / CPR, k = k0 + (c % CPR) * EPC;
      ok = (row < dim) && (k < Kred);
      off = (int64_t)row * ld + k;
    } else {
      constexpr int CPR = BM / EPC;
      const int r = k0 + c / CPR, col = tile0 + (c % CPR) * EPC;
      ok = (r < Kred) && (col < dim);
      off = (int64_t)r * ld + col;
    }
    regs[i] = ok ? *reinterpret_cast<const uint4*>(base + off) : make_uint4(0, 0, 0, 0);
  }
}

// ---- registers -> LDS (swizzled) -------------------------------------------------------------
template <typename T, bool TR, int BK, int NCH>
__device__ __forceinline__ void stage_store(unsigned char* s, const uint4 (&regs)[NCH], int tid) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int ROWB = BK * (int)sizeof(T);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTHR;
    int off;
    if (!TR) {
      constexpr int CPR = BK / EPC;
      const int row = c / CPR, kc = c % CPR;
      off = row * ROWB + ((kc ^ swz_n<ROWB>(row)) << 4);
    } else {
      constexpr int CPR = BM / EPC;
      const int r = c / CPR, cc = c % CPR;
      if (sizeof(T) == 2) off = r * 256 + ((((cc >> 1) ^ swz_t(r))) << 5) + ((cc & 1) << 4);
      else off = r * 512 + (cc << 4);
    }
    *reinterpret_cast<uint4*>(s + off) = regs[i];
  }
}

// ---- LDS -> MFMA fragments -------------------------------------------------------------------
// bf16: the 32-deep MFMA k-step `hh` of a BK-deep tile
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

template <typename T, bool ATR, bool BTR, int BK, bool BSUM>
__device__ __forceinline__ void tile_mma(const unsigned char* sA, const unsigned char* sB, f32x4_t (&acc)[4][4], f32x4_t (&accb)[4],
                                         bool do_bsum, int wm, int wn, int lane) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int hh = 0; hh < BK / 32; ++hh) {
      bf16x8_t af[4], bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = frag_bf16<ATR, BK>(sA, wm * 64 + t * 16, lane, hh);
#pragma unroll
      for (int t = 0; t < 4; ++t) bf[t] = frag_bf16<BTR, BK>(sB, wn * 64 + t * 16, lane, hh);
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tj], af[ti], acc[ti][tj], 0, 0, 0);
      if constexpr (BSUM) {
        if (do_bsum) {          // wave-uniform: column sums of the A tile = A^T * ones
          typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
          const u16x8_t o16 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
          const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, o16);
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) accb[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[ti], accb[ti], 0, 0, 0);
        }
      }
    }
  } else {
    static_assert(sizeof(T) == 2 || BK == 32, "fp32 tiles are 32 deep");
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float af[4][4], bf[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) frag_f32<ATR>(sA, wm * 64 + t * 16, lane, hh, af[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) frag_f32<BTR>(sB, wn * 64 + t * 16, lane, hh, bf[t]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tj][q], af[ti][q], acc[ti][tj], 0, 0, 0);
        if constexpr (BSUM) {
          if (do_bsum) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) accb[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, af[ti][q], accb[ti], 0, 0, 0);
          }
        }
      }
    }
  }
}

// ---- coalesced epilogue: fp32 tile -> swizzled LDS (64 rows per pass) -> 16-byte row-major stores --------------
// PRE = true : out_pre = acc + bias                      (pre-activation copy kept for the backward pass)
// PRE = false: out = res + row_scale * act(acc + bias)   (act = GELU here; GELU' (x aux) applied in phase B)
template <typename T, bool PRE>
__device__ __forceinline__ void store_tile(unsigned char* smem, const f32x4_t (&acc)[4][4], const Problem& P, int act, int N, int64_t ldc,
                                           int m0, int n0, int wm, int wn, int lane, int tid) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int CPR = BN / EPC;                 // output chunks per tile row
  float* sT = reinterpret_cast<float*>(smem);   // [64][128] fp32, float4 chunk c of row r stored at c ^ (r & 7)
  T* outp = reinterpret_cast<T*>(PRE ? P.out_pre : P.out);
#pragma unroll 1
  for (int p = 0; p < 2; ++p) {
    __syncthreads();
    if (wm == p) {
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int n = n0 + wn * 64 + tj * 16 + (lane >> 4) * 4;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.bias && n < N) b4 = *reinterpret_cast<const float4*>(P.bias + n);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          float4 v = make_float4(acc[ti][tj][0] + b4.x, acc[ti][tj][1] + b4.y, acc[ti][tj][2] + b4.z, acc[ti][tj][3] + b4.w);
          if (!PRE && act == LMV_ACT_GELU) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
          const int r = ti * 16 + (lane & 15), c4 = wn * 16 + tj * 4 + (lane >> 4);
          *reinterpret_cast<float4*>(sT + r * BN + ((c4 ^ (r & 7)) << 2)) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 64 * CPR / NTHR; ++i) {
      const int c = tid + i * NTHR, r = c / CPR, oc = c % CPR;
      const int m = m0 + p * 64 + r, n = n0 + oc * EPC;
      if (m >= P.M || n >= N) continue;
      float v[EPC];
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        const int c4 = (oc * EPC + e) >> 2;
        const float4 t = *reinterpret_cast<const float4*>(sT + r * BN + ((c4 ^ (r & 7)) << 2));
        v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
      }
      const int64_t o = (int64_t)m * ldc + n;
      if (!PRE) {
        if (act == LMV_ACT_GELU_GRAD) {
          float u[EPC];
          chunk_to_f<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(P.aux) + o), u);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] *= gelu_grad_f(u[e]);
        }
        if (P.row_scale) {
          const float rs = P.row_scale[m / P.rps];
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] *= rs;
        }
        if (P.res) {
          float r8[EPC];
          chunk_to_f<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(P.res) + o), r8);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] += r8[e];
        }
      }
      *reinterpret_cast<uint4*>(outp + o) = f_to_chunk<T>(v);
    }
  }
}

// ---- global -> LDS directly (bf16): global_load_lds_dwordx4 writes  wave-uniform base + lane * 16, i.e. each
// wave-instruction fills one LINEAR 1 KiB segment of the tile image.  The XOR swizzle therefore goes on the per-lane
// SOURCE address: the lane that lands on physical chunk p of row r fetches logical chunk p ^ swz(r).  No staging
// VGPRs and no ds_write pass (ds_write_b128 costs ~13 LDS cycles per wave-instruction and made the register-staged
// loop LDS-bound).  Lanes whose tile row / column lies outside the matrix are masked off: their LDS bytes stay stale,
// which only ever feeds accumulators of outputs that are not stored (the reduction dimension is never masked here:
// the host uses this path only when Kred is a multiple of BK).
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

template <bool TR, int BK>
__device__ __forceinline__ void stage_dma(unsigned char* tile, const bf16_t* __restrict__ base, int64_t ld, int dim, int tile0, int k0,
                                          int lane, int wave) {
  constexpr int NSEG = BM * BK * 2 / 1024;          // 1 KiB segments per tile image
#pragma unroll
  for (int i = 0; i < NSEG / 4; ++i) {
    const int sg = wave + i * 4;
    const bf16_t* src;
    bool ok;
    // out-of-range rows / columns are CLAMPED, not masked: every wave issues exactly the same number of LDS-DMA
    // instructions per tile (the ring pipeline below counts them with s_waitcnt vmcnt(N)), and the duplicated data
    // only reaches accumulators of outputs that are never stored.
    if (!TR) {
      constexpr int ROWB = BK * 2, CPR = ROWB / 16, RPS = 1024 / ROWB;
      const int r = sg * RPS + lane / CPR, p = lane % CPR, kc = p ^ swz_n<ROWB>(r);
      src = base + (int64_t)min(tile0 + r, dim - 1) * ld + k0 + kc * 8;
    } else {
      const int r = sg * 4 + (lane >> 4), p16 = lane & 15, c16 = ((((p16 >> 1) ^ swz_t(r))) << 1) | (p16 & 1);
      src = base + (int64_t)(k0 + r) * ld + min(tile0 + c16 * 8, dim - 8);
    }
    (void)ok;
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(tile + sg * 1024), 16, 0, 0);
  }
}

// wait until at most `pending` whole tiles (DPT LDS-DMA instructions each) of this wave are still in flight
template <int DPT>
__device__ __forceinline__ void wait_tiles(int pending) {
  if (pending <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (pending == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPT) : "memory");
  else if (pending == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPT) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * DPT) : "memory");
}

template <typename T, bool ATR, bool BTR, bool SPLITK, int BK, bool DMA, int NBUF>
__global__ __launch_bounds__(NTHR) void gemm_kernel(const GemmArgs g) {
  constexpr int NCH = (BM * BK * (int)sizeof(T)) / 16 / NTHR;
  constexpr int TILE_BYTES = BM * BK * (int)sizeof(T);
  static_assert(DMA || NBUF == 2, "the register-staged loop is double-buffered");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * 2 * TILE_BYTES];   // [buffer][A | B]; reused by the epilogue
  static_assert(NBUF * 2 * TILE_BYTES >= 64 * BN * 4, "epilogue staging needs 32 KB");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: hardware places block b on XCD b % 8; give each XCD a contiguous run of logical tiles so
  // the n-tiles of one m-tile share an L2 (speed only -- any placement is correct).
  int bid;
  {
    const int T_ = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = T_ >> 3, r = T_ & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int pi = (g.nprob > 1 && bid >= g.p[1].tile_begin) ? 1 : 0;
  const Problem& P = g.p[pi];
  bid -= P.tile_begin;
  const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = P.M, N = g.N, Kred = P.Kred;
  const int kt_total = (Kred + BK - 1) / BK;
  const int kt_beg = blockIdx.y * g.kt_per_split;
  const int kt_end = min(kt_total, kt_beg + g.kt_per_split);
  if (kt_beg >= kt_end) return;

  const T* __restrict__ A = reinterpret_cast<const T*>(P.a);
  const T* __restrict__ Bw = reinterpret_cast<const T*>(P.b);

  f32x4_t acc[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const bool do_bsum = SPLITK && (P.bias_grad != nullptr) && tn == 0 && wn == 0;      // wave-uniform
  int cur = 0;
  if constexpr (DMA) {
    static_assert(sizeof(T) == 2, "the direct-to-LDS path is bf16 only");
    const bf16_t* A16 = reinterpret_cast<const bf16_t*>(A);
    const bf16_t* B16 = reinterpret_cast<const bf16_t*>(Bw);
    // Ring of NBUF tile buffers with up to NBUF-1 k-tiles in flight.  HBM latency (~2 us under load) is far longer
    // than one tile's MFMAs, so a single prefetched tile leaves the matrix cores idle; LDS-DMA requests are kept in
    // flight ACROSS barriers by counting them (s_waitcnt vmcnt(N), raw s_barrier -- a __syncthreads() would drain
    // them).  Per k-tile: wait for tile kt (own DMAs) -> barrier (everyone's DMAs for kt landed, everyone finished
    // reading the buffer of tile kt-1) -> refill that buffer with tile kt+NBUF-1 -> MFMAs on tile kt.
    constexpr int DPT = 2 * (BM * BK * 2 / 1024) / 4;   // LDS-DMA instructions per wave per k-tile (A + B)
    const int nk = kt_end - kt_beg;
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) {
      if (s < nk) {
        stage_dma<ATR, BK>(smem + s * 2 * TILE_BYTES, A16, g.lda, M, m0, (kt_beg + s) * BK, lane, wave);
        stage_dma<BTR, BK>(smem + s * 2 * TILE_BYTES + TILE_BYTES, B16, g.ldb, N, n0, (kt_beg + s) * BK, lane, wave);
      }
    }
    for (int i = 0; i < nk; ++i) {
      const int issued_after = min(NBUF - 2, nk - 1 - i);          // tiles issued after tile i and still allowed in flight
      wait_tiles<DPT>(issued_after);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (i + NBUF - 1 < nk) {
        unsigned char* dA = smem + ((i + NBUF - 1) % NBUF) * 2 * TILE_BYTES;
        stage_dma<ATR, BK>(dA, A16, g.lda, M, m0, (kt_beg + i + NBUF - 1) * BK, lane, wave);
        stage_dma<BTR, BK>(dA + TILE_BYTES, B16, g.ldb, N, n0, (kt_beg + i + NBUF - 1) * BK, lane, wave);
      }
      const unsigned char* sA = smem + (i % NBUF) * 2 * TILE_BYTES;
      tile_mma<T, ATR, BTR, BK, SPLITK>(sA, sA + TILE_BYTES, acc, accb, do_bsum, wm, wn, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's LDS reads of tile i are complete
    }
    __syncthreads();
  } else {
    uint4 ra[NCH], rb[NCH];
    stage_load<T, ATR, BK, NCH>(ra, A, g.lda, M, Kred, m0, kt_beg * BK, tid);
    stage_load<T, BTR, BK, NCH>(rb, Bw, g.ldb, N, Kred, n0, kt_beg * BK, tid);
    stage_store<T, ATR, BK, NCH>(smem, ra, tid);
    stage_store<T, BTR, BK, NCH>(smem + TILE_BYTES, rb, tid);
    __syncthreads();
    for (int kt = kt_beg; kt < kt_end; ++kt) {
      const bool has_next = kt + 1 < kt_end;
      if (has_next) {                                   // global loads of the next k-tile fly under this tile's MFMAs
        stage_load<T, ATR, BK, NCH>(ra, A, g.lda, M, Kred, m0, (kt + 1) * BK, tid);
        stage_load<T, BTR, BK, NCH>(rb, Bw, g.ldb, N, Kred, n0, (kt + 1) * BK, tid);
      }
      const unsigned char* sA = smem + cur * 2 * TILE_BYTES;
      const unsigned char* sB = sA + TILE_BYTES;
      tile_mma<T, ATR, BTR, BK, SPLITK>(sA, sB, acc, accb, do_bsum, wm, wn, lane);
      if (has_next) {
        unsigned char* dA = smem + (cur ^ 1) * 2 * TILE_BYTES;
        stage_store<T, ATR, BK, NCH>(dA, ra, tid);
        stage_store<T, BTR, BK, NCH>(dA + TILE_BYTES, rb, tid);
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  if constexpr (SPLITK) {
    // partial tile -> this split's slab (plain stores; summed by splitk_reduce_kernel)
    float* slab = g.ws + (int64_t)(g.slab_base[pi] + blockIdx.y) * g.slab_stride;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int m = m0 + wm * 64 + ti * 16 + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int n = n0 + wn * 64 + tj * 16 + (lane >> 4) * 4;
        if (n >= N) continue;
        *reinterpret_cast<float4*>(slab + (int64_t)m * g.ldc + n) = make_float4(acc[ti][tj][0], acc[ti][tj][1], acc[ti][tj][2], acc[ti][tj][3]);
      }
      if (do_bsum && lane < 16) slab[(int64_t)M * g.ldc + m] = accb[ti][0];
    }
  } else {
    if (P.out_pre) store_tile<T, true>(smem, acc, P, g.act, N, g.ldc, m0, n0, wm, wn, lane, tid);
    store_tile<T, false>(smem, acc, P, g.act, N, g.ldc, m0, n0, wm, wn, lane, tid);
  }
}

// out[i] += sum_s ws[s][i]  (i < nw: dW; nw <= i < nw + nb: db)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int nslabs, int64_t stride, float* __restrict__ out_w,
                                                           int64_t nw, float* __restrict__ out_b, int nb) {
  const int64_t n4 = (nw + (out_b ? nb : 0)) >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = ws + i * 4;
#pragma unroll 4
    for (int s = 0; s < nslabs; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)s * stride);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float* o = (i * 4 < nw) ? out_w + i * 4 : out_b + (i * 4 - nw);
    float4 c = *reinterpret_cast<float4*>(o);
    c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w;
    *reinterpret_cast<float4*>(o) = c;
  }
}

enum Mode { MODE_FWD = 0, MODE_DX = 1, MODE_DW = 2 };

struct Plan { GemmArgs g; int total, splits, bk, nbuf, nsplit[2]; size_t ws_bytes; };

int make_plan(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, Mode mode, Plan* pl) {
  if (nproblems < 1 || nproblems > 2) LMV_FAIL(LMV_ERR_SHAPE, "linear: nproblems must be 1 or 2 (got %d)", nproblems);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "linear: unsupported dtype %d", dtype);
  if (N <= 0 || K <= 0 || (N % 8) || (K % 8)) LMV_FAIL(LMV_ERR_SHAPE, "linear: N=%d, K=%d must be positive multiples of 8", N, K);
  GemmArgs& g = pl->g;
  g = GemmArgs{};
  g.nprob = nproblems; g.act = act;
  int out_cols;
  switch (mode) {
    case MODE_FWD: g.N = N; g.lda = K; g.ldb = K; g.ldc = N; out_cols = N; break;
    case MODE_DX:  g.N = K; g.lda = N; g.ldb = K; g.ldc = K; out_cols = K; break;
    default:       g.N = K; g.lda = N; g.ldb = K; g.ldc = K; out_cols = K; break;
  }
  g.tiles_n = (out_cols + BN - 1) / BN;
  int total = 0;
  for (int i = 0; i < nproblems; ++i) {
    const lmv_linear_problem& q = p[i];
    if (q.rows <= 0 || q.rows > 0x7fffffffLL / 4) LMV_FAIL(LMV_ERR_SHAPE, "linear: bad rows %lld", (long long)q.rows);
    if (!q.a || !q.w || !q.out) LMV_FAIL(LMV_ERR_SHAPE, "linear: null operand");
    if (!lmv_aligned16(q.a) || !lmv_aligned16(q.w) || !lmv_aligned16(q.out) || !lmv_aligned16(q.res) || !lmv_aligned16(q.aux) ||
        !lmv_aligned16(q.out_pre) || !lmv_aligned16(q.bias) || !lmv_aligned16(q.bias_grad))
      LMV_FAIL(LMV_ERR_SHAPE, "linear: operands must be 16-byte aligned");
    if (q.row_scale && q.rows_per_sample <= 0) LMV_FAIL(LMV_ERR_SHAPE, "linear: rows_per_sample must be > 0 with row_scale");
    Problem& P = g.p[i];
    P.a = q.a; P.b = q.w; P.bias = q.bias; P.res = q.res; P.row_scale = q.row_scale; P.aux = q.aux;
    P.out = q.out; P.out_pre = q.out_pre; P.bias_grad = q.bias_grad; P.rps = q.rows_per_sample > 0 ? q.rows_per_sample : 1;
    if (mode == MODE_DW) { P.M = N; P.Kred = (int)q.rows; }
    else { P.M = (int)q.rows; P.Kred = (mode == MODE_FWD) ? K : N; }
    P.tiles_m = (P.M + BM - 1) / BM;
    P.tile_begin = total;
    total += P.tiles_m * g.tiles_n;
  }
  const bool bf = dtype == LMV_BF16;
  // bf16: 64-deep k-tiles unless the reduction is a short non-multiple of 64 (C = 96 layers); fp32: 32-deep
  int min_kred = 1 << 30;
  bool all64 = true;
  for (int i = 0; i < nproblems; ++i) { if (g.p[i].Kred < min_kred) min_kred = g.p[i].Kred; if (g.p[i].Kred % 64) all64 = false; }
  // LMV_GEMM_RING=0: 64-deep tiles, double-buffered (2 x 32 KB); default: 32-deep tiles in a 4-slot LDS-DMA ring (4 x 16 KB,
  // up to 3 k-tiles in flight per workgroup)
  static const bool ring = [] { const char* e = getenv("LMV_GEMM_RING"); return !e || atoi(e) != 0; }();
  static const bool no_dma_env = getenv("LMV_GEMM_NO_DMA") != nullptr;
  bool all32 = true;
  for (int i = 0; i < nproblems; ++i) if (g.p[i].Kred % 32) all32 = false;
  const bool use_ring = bf && ring && !no_dma_env && all32;
  const int bk = use_ring ? 32 : ((bf && (mode == MODE_DW || all64 || min_kred >= 512)) ? 64 : 32);
  pl->nbuf = use_ring ? 4 : 2;
  int max_kt = 1;
  for (int i = 0; i < nproblems; ++i) { const int kt = (g.p[i].Kred + bk - 1) / bk; if (kt > max_kt) max_kt = kt; }
  int splits = 1;
  g.kt_per_split = max_kt;
  pl->ws_bytes = 0;
  pl->nsplit[0] = pl->nsplit[1] = 1;
  if (mode == MODE_DW) {
    static const int target = [] { const char* e = getenv("LMV_DW_TARGET_BLOCKS"); return e ? atoi(e) : 768; }();
    splits = (target + total - 1) / total;
    const int max_splits = (max_kt + 7) / 8;   // at least 8 k-tiles per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    g.kt_per_split = (max_kt + splits - 1) / splits;
    splits = (max_kt + g.kt_per_split - 1) / g.kt_per_split;
    int slabs = 0;
    for (int i = 0; i < nproblems; ++i) {
      const int kt = (g.p[i].Kred + bk - 1) / bk;
      pl->nsplit[i] = (kt + g.kt_per_split - 1) / g.kt_per_split;
      g.slab_base[i] = slabs;
      slabs += pl->nsplit[i];
    }
    g.slab_stride = (int64_t)N * K + N;
    pl->ws_bytes = (size_t)slabs * g.slab_stride * sizeof(float);
  }
  pl->total = total; pl->splits = splits; pl->bk = bk;
  return LMV_OK;
}

int launch(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream, Mode mode, void* ws, size_t ws_bytes) {
  Plan pl;
  if (int rc = make_plan(p, nproblems, N, K, act, dtype, mode, &pl)) return rc;
  GemmArgs& g = pl.g;
  if (mode == MODE_DW) {
    if (!ws || ws_bytes < pl.ws_bytes || !lmv_aligned16(ws)) LMV_FAIL(LMV_ERR_WORKSPACE, "linear_dw: workspace %zu < %zu bytes", ws_bytes, pl.ws_bytes);
    g.ws = reinterpret_cast<float*>(ws);
  }
  const bool bf = dtype == LMV_BF16;
  const int bk = pl.bk;
  dim3 grid(pl.total, pl.splits), block(NTHR);
  hipStream_t st = (hipStream_t)stream;
  // direct-to-LDS staging needs an unmasked reduction dimension (and is bf16 only); LMV_GEMM_NO_DMA=1 forces the
  // register-staged kernels (A/B testing)
  static const bool no_dma = getenv("LMV_GEMM_NO_DMA") != nullptr;
  bool dma = bf && !no_dma;
  for (int i = 0; i < nproblems; ++i) if (g.p[i].Kred % bk) dma = false;
#define LAUNCH(T, A, B, SK, KK, DM, NB) hipLaunchKernelGGL((gemm_kernel<T, A, B, SK, KK, DM, NB>), grid, block, 0, st, g)
#define LAUNCH_BF(A, B, SK)                                                                                           \
  do {                                                                                                                \
    if (dma && pl.nbuf == 4) LAUNCH(bf16_t, A, B, SK, 32, true, 4);                                                    \
    else if (bk == 64) { if (dma) LAUNCH(bf16_t, A, B, SK, 64, true, 2); else LAUNCH(bf16_t, A, B, SK, 64, false, 2); } \
    else               { if (dma) LAUNCH(bf16_t, A, B, SK, 32, true, 2); else LAUNCH(bf16_t, A, B, SK, 32, false, 2); } \
  } while (0)
  if (mode == MODE_FWD) {
    if (!bf) LAUNCH(float, false, false, false, 32, false, 2); else LAUNCH_BF(false, false, false);
  } else if (mode == MODE_DX) {
    if (!bf) LAUNCH(float, false, true, false, 32, false, 2); else LAUNCH_BF(false, true, false);
  } else {
    if (!bf) LAUNCH(float, true, true, true, 32, false, 2); else LAUNCH_BF(true, true, true);
  }
#undef LAUNCH_BF
#undef LAUNCH
  LMV_CHECK_LAUNCH("linear");
  if (mode == MODE_DW) {
    const int64_t nw = (int64_t)N * K;
    const bool shared = nproblems == 2 && p[0].out == p[1].out;
    for (int i = 0; i < nproblems; ++i) {
      if (shared && i == 1) break;
      const int nslabs = shared ? pl.nsplit[0] + pl.nsplit[1] : pl.nsplit[i];
      if (shared && p[0].bias_grad != p[1].bias_grad) LMV_FAIL(LMV_ERR_SHAPE, "linear_dw: problems sharing dW must share db");
      const float* base = g.ws + (int64_t)g.slab_base[i] * g.slab_stride;
      int blocks = (int)((nw / 4 + 255) / 256); if (blocks > 1024) blocks = 1024;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, base, nslabs, g.slab_stride, reinterpret_cast<float*>(p[i].out), nw,
                         p[i].bias_grad, N);
    }
    LMV_CHECK_LAUNCH("linear_dw reduce");
  }
  return LMV_OK;
}

}  // namespace

extern "C" int lmv_linear_fwd(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream) {
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU) LMV_FAIL(LMV_ERR_SHAPE, "linear_fwd: act must be NONE or GELU");
  return launch(p, nproblems, N, K, act, dtype, stream, MODE_FWD, nullptr, 0);
}
extern "C" int lmv_linear_dx(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream) {
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU_GRAD) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx: act must be NONE or GELU_GRAD");
  for (int i = 0; i < nproblems && i < 2; ++i)
    if (act == LMV_ACT_GELU_GRAD && !p[i].aux) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx: GELU_GRAD needs aux");
  return launch(p, nproblems, N, K, act, dtype, stream, MODE_DX, nullptr, 0);
}
extern "C" size_t lmv_linear_dw_workspace_bytes(const lmv_linear_problem* p, int nproblems, int N, int K, int dtype) {
  Plan pl;
  if (make_plan(p, nproblems, N, K, LMV_ACT_NONE, dtype, MODE_DW, &pl)) return 0;
  return pl.ws_bytes;
}
extern "C" int lmv_linear_dw(const lmv_linear_problem* p, int nproblems, int N, int K, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  return launch(p, nproblems, N, K, LMV_ACT_NONE, dtype, stream, MODE_DW, workspace, workspace_bytes);
}
