// gemm.hip -- MFMA GEMM for the Linear layers of the LeMeViT blocks (gfx950).
//
// One kernel template covers the three contractions a Linear layer needs:
//   fwd  C[r,n] = sum_k  A[r,k]  W[n,k]      (A, W both reduction-contiguous: plain 16-B LDS reads)
//   dx   C[r,k] = sum_n dY[r,n]  W[n,k]      (W is reduction-MAJOR: ds_read_b64_tr_b16 transpose reads)
//   dw   C[n,k] = sum_r dY[r,n]  X[r,k]      (both operands reduction-major, split over r, fp32 atomics)
// so the backward pass needs neither transposed weight copies nor transposed activations.
//
// Tile: 128 x 128 x 32 per 256-thread workgroup, 4 waves as 2 x 2, each wave 64 x 64 =
// 4 x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16, or v_mfma_f32_16x16x4_f32 in the exact-fp32 mode).
// Operands are register-staged (global -> VGPR -> LDS) with the next k-tile's loads in flight under
// the current tile's MFMAs.  LDS images are XOR-swizzled so that both the 16-B fragment reads and
// the transpose reads are bank-conflict free (derivation in DESIGN.md).
// The MFMA operands are passed swapped (W as "A") so every lane ends up with 4 CONSECUTIVE output
// columns of one row: the epilogue (bias, GELU / GELU', DropPath scale, residual) runs on 8/16-byte
// vectors.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NTHR = 256;

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
};

template <typename T> __device__ __forceinline__ int swz_n(int row);
template <> __device__ __forceinline__ int swz_n<bf16_t>(int row) { return (-(row >> 2)) & 3; }
template <> __device__ __forceinline__ int swz_n<float>(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_t(int r) { return (r & 3) | ((r >> 1) & 4); }

// ---- global -> registers ---------------------------------------------------------------------
// Normal operand: tile [128 rows][BK] (reduction-contiguous).  TR operand: tile [BK rows][128 cols].
template <typename T, bool TR, int NCH>
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
template <typename T, bool TR, int NCH>
__device__ __forceinline__ void stage_store(unsigned char* s, const uint4 (&regs)[NCH], int tid) {
  constexpr int EPC = DT<T>::EPC;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTHR;
    int off;
    if (!TR) {
      constexpr int CPR = BK / EPC;
      const int row = c / CPR, kc = c % CPR;
      off = row * (BK * (int)sizeof(T)) + ((kc ^ swz_n<T>(row)) << 4);
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
template <bool TR>
__device__ __forceinline__ bf16x8_t frag_bf16(const unsigned char* s, int base, int lane) {
  if (!TR) {
    const int row = base + (lane & 15), kc = lane >> 4;
    return *reinterpret_cast<const bf16x8_t*>(s + row * 64 + ((kc ^ swz_n<bf16_t>(row)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, rr = i >> 2, q = i & 3;
    const int r1 = g * 8 + rr, r2 = r1 + 4, ch = base >> 4;
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
    const float4 v = *reinterpret_cast<const float4*>(s + row * 128 + ((kc ^ swz_n<float>(row)) << 4));
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    const int col = base + (lane & 15), r0 = hh * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const float*>(s + (r0 + q) * 512 + col * 4);
  }
}

template <typename T, bool ATR, bool BTR>
__device__ __forceinline__ void tile_mma(const unsigned char* sA, const unsigned char* sB, f32x4_t (&acc)[4][4], int wm, int wn, int lane) {
  if constexpr (sizeof(T) == 2) {
    bf16x8_t af[4], bf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) af[t] = frag_bf16<ATR>(sA, wm * 64 + t * 16, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) bf[t] = frag_bf16<BTR>(sB, wn * 64 + t * 16, lane);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tj], af[ti], acc[ti][tj], 0, 0, 0);
  } else {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float af[4][4], bf[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) frag_f32<ATR>(sA, wm * 64 + t * 16, lane, hh, af[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) frag_f32<BTR>(sB, wn * 64 + t * 16, lane, hh, bf[t]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tj][q], af[ti][q], acc[ti][tj], 0, 0, 0);
    }
  }
}

template <typename T>
__device__ __forceinline__ float tr_colsum(const unsigned char* s, int c) {
  float a = 0.f;
#pragma unroll 8
  for (int r = 0; r < BK; ++r) {
    if (sizeof(T) == 2) a += bf2f(*reinterpret_cast<const bf16_t*>(s + r * 256 + (((c >> 4) ^ swz_t(r)) << 5) + (c & 15) * 2));
    else a += *reinterpret_cast<const float*>(s + r * 512 + c * 4);
  }
  return a;
}

template <typename T, bool ATR, bool BTR, bool ATOMIC>
__global__ __launch_bounds__(NTHR) void gemm_kernel(const GemmArgs g) {
  constexpr int NCH = (BM * BK * (int)sizeof(T)) / 16 / NTHR;
  constexpr int TILE_BYTES = BM * BK * (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
  unsigned char* sA = smem;
  unsigned char* sB = smem + TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
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

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[NCH], rb[NCH];
  stage_load<T, ATR, NCH>(ra, A, g.lda, M, Kred, m0, kt_beg * BK, tid);
  stage_load<T, BTR, NCH>(rb, Bw, g.ldb, N, Kred, n0, kt_beg * BK, tid);
  const bool do_bsum = ATR && (P.bias_grad != nullptr) && tn == 0 && tid < BM;
  float bsum = 0.f;

  for (int kt = kt_beg; kt < kt_end; ++kt) {
    stage_store<T, ATR, NCH>(sA, ra, tid);
    stage_store<T, BTR, NCH>(sB, rb, tid);
    __syncthreads();
    if (kt + 1 < kt_end) {
      stage_load<T, ATR, NCH>(ra, A, g.lda, M, Kred, m0, (kt + 1) * BK, tid);
      stage_load<T, BTR, NCH>(rb, Bw, g.ldb, N, Kred, n0, (kt + 1) * BK, tid);
    }
    tile_mma<T, ATR, BTR>(sA, sB, acc, wm, wn, lane);
    if (ATR) { if (do_bsum) bsum += tr_colsum<T>(sA, tid); }
    __syncthreads();
  }

  // ---- epilogue: lane owns row m, columns n .. n+3 of each 16x16 tile -----------------------
  const int64_t ldc = g.ldc;
  const int act = g.act;
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    const int m = m0 + wm * 64 + ti * 16 + (lane & 15);
    if (m >= M) continue;
    float rs = 1.f;
    if (!ATOMIC && P.row_scale) rs = P.row_scale[m / P.rps];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
      const int n = n0 + wn * 64 + tj * 16 + (lane >> 4) * 4;
      if (n >= N) continue;
      float v[4] = {acc[ti][tj][0], acc[ti][tj][1], acc[ti][tj][2], acc[ti][tj][3]};
      const int64_t o = (int64_t)m * ldc + n;
      if (ATOMIC) {
        float* out = reinterpret_cast<float*>(P.out) + o;
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(out + q, v[q]);
        continue;
      }
      if (P.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(P.bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      if (P.out_pre) st4(reinterpret_cast<T*>(P.out_pre) + o, v);
      if (act == LMV_ACT_GELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gelu_f(v[q]);
      } else if (act == LMV_ACT_GELU_GRAD) {
        float u[4];
        ld4(reinterpret_cast<const T*>(P.aux) + o, u);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= gelu_grad_f(u[q]);
      }
      if (P.row_scale) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= rs;
      }
      if (P.res) {
        float r4[4];
        ld4(reinterpret_cast<const T*>(P.res) + o, r4);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += r4[q];
      }
      st4(reinterpret_cast<T*>(P.out) + o, v);
    }
  }
  if (ATR) {
    if (do_bsum && m0 + tid < M) atomicAdd(P.bias_grad + m0 + tid, bsum);
  }
}

enum Mode { MODE_FWD = 0, MODE_DX = 1, MODE_DW = 2 };

int launch(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream, Mode mode) {
  if (nproblems < 1 || nproblems > 2) LMV_FAIL(LMV_ERR_SHAPE, "linear: nproblems must be 1 or 2 (got %d)", nproblems);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "linear: unsupported dtype %d", dtype);
  if (N <= 0 || K <= 0 || (N % 8) || (K % 8)) LMV_FAIL(LMV_ERR_SHAPE, "linear: N=%d, K=%d must be positive multiples of 8", N, K);
  GemmArgs g{};
  g.nprob = nproblems; g.act = act;
  int out_cols;
  switch (mode) {
    case MODE_FWD: g.N = N; g.lda = K; g.ldb = K; g.ldc = N; out_cols = N; break;
    case MODE_DX:  g.N = K; g.lda = N; g.ldb = K; g.ldc = K; out_cols = K; break;
    default:       g.N = K; g.lda = N; g.ldb = K; g.ldc = K; out_cols = K; break;
  }
  g.tiles_n = (out_cols + BN - 1) / BN;
  int total = 0, max_kt = 1;
  for (int i = 0; i < nproblems; ++i) {
    const lmv_linear_problem& q = p[i];
    if (q.rows <= 0 || q.rows > 0x7fffffffLL / 4) LMV_FAIL(LMV_ERR_SHAPE, "linear: bad rows %lld", (long long)q.rows);
    if (!q.a || !q.w || !q.out) LMV_FAIL(LMV_ERR_SHAPE, "linear: null operand");
    if (!lmv_aligned16(q.a) || !lmv_aligned16(q.w) || !lmv_aligned16(q.out) || !lmv_aligned16(q.res) || !lmv_aligned16(q.aux) ||
        !lmv_aligned16(q.out_pre) || !lmv_aligned16(q.bias))
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
    const int kt = (P.Kred + BK - 1) / BK;
    if (kt > max_kt) max_kt = kt;
  }
  int splits = 1;
  g.kt_per_split = max_kt;
  if (mode == MODE_DW) {
    // split the token reduction so the launch fills the chip (>= ~4 workgroups per CU)
    splits = (1024 + total - 1) / total;
    const int max_splits = (max_kt + 7) / 8;   // at least 8 k-tiles (256 rows) per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    g.kt_per_split = (max_kt + splits - 1) / splits;
    splits = (max_kt + g.kt_per_split - 1) / g.kt_per_split;
  }
  dim3 grid(total, splits), block(NTHR);
  hipStream_t st = (hipStream_t)stream;
  const bool bf = dtype == LMV_BF16;
  if (mode == MODE_FWD) {
    if (bf) hipLaunchKernelGGL((gemm_kernel<bf16_t, false, false, false>), grid, block, 0, st, g);
    else    hipLaunchKernelGGL((gemm_kernel<float, false, false, false>), grid, block, 0, st, g);
  } else if (mode == MODE_DX) {
    if (bf) hipLaunchKernelGGL((gemm_kernel<bf16_t, false, true, false>), grid, block, 0, st, g);
    else    hipLaunchKernelGGL((gemm_kernel<float, false, true, false>), grid, block, 0, st, g);
  } else {
    if (bf) hipLaunchKernelGGL((gemm_kernel<bf16_t, true, true, true>), grid, block, 0, st, g);
    else    hipLaunchKernelGGL((gemm_kernel<float, true, true, true>), grid, block, 0, st, g);
  }
  LMV_CHECK_LAUNCH("linear");
  return LMV_OK;
}

}  // namespace

extern "C" int lmv_linear_fwd(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream) {
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU) LMV_FAIL(LMV_ERR_SHAPE, "linear_fwd: act must be NONE or GELU");
  return launch(p, nproblems, N, K, act, dtype, stream, MODE_FWD);
}
extern "C" int lmv_linear_dx(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream) {
  if (act != LMV_ACT_NONE && act != LMV_ACT_GELU_GRAD) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx: act must be NONE or GELU_GRAD");
  for (int i = 0; i < nproblems && i < 2; ++i)
    if (act == LMV_ACT_GELU_GRAD && !p[i].aux) LMV_FAIL(LMV_ERR_SHAPE, "linear_dx: GELU_GRAD needs aux");
  return launch(p, nproblems, N, K, act, dtype, stream, MODE_DX);
}
extern "C" int lmv_linear_dw(const lmv_linear_problem* p, int nproblems, int N, int K, int dtype, void* stream) {
  return launch(p, nproblems, N, K, LMV_ACT_NONE, dtype, stream, MODE_DW);
}
