// attn.hip -- attention cores of the LeMeViT blocks, head dim 32, exact-softmax, fp32 math.
//
// Three shape regimes occur in the backbone:
//   many queries x few keys   image tokens attend to the 16 meta tokens      (DCA x-direction)
//   few queries  x many keys  16 meta tokens attend to N image tokens        (DCA c-direction, stage-0 CA)
//   L x L self attention      late stages, L = 196 / 49 (and 16 x 16 for the meta tokens)
// Operands are read IN PLACE from the packed projections through (batch, row) strides.
//
// Round-1 kernels are wavefront-parallel VALU kernels shared by the fp32 and bf16 modes
// (query-per-lane or key-per-lane, K/V or Q/dO tiles broadcast from LDS, online softmax in
// registers, LDS-staged small GEMMs for the 16-row reductions).  The bf16 MFMA variants replace
// the hot ones later ("attention roadmap").
#include "common.h"
#include "attn_internal.h"

namespace {

constexpr int D = 32;
constexpr int QB = 128;   // queries (or keys) per workgroup in the lane-per-row kernels
constexpr int KC = 64;    // rows per LDS chunk
constexpr int FQ = 16;    // max queries of the few-query path
constexpr int FKB = 256;  // keys per workgroup in the few-query forward

typedef AttnArgs Args;

template <typename T>
__device__ __forceinline__ void load_row(const T* p, float* f) {   // 32 contiguous elements -> floats
  constexpr int EPC = DT<T>::EPC;
#pragma unroll
  for (int c = 0; c < D / EPC; ++c) chunk_to_f<T>(*reinterpret_cast<const uint4*>(p + c * EPC), f + c * EPC);
}
template <typename T>
__device__ __forceinline__ void store_row(T* p, const float* f) {
  constexpr int EPC = DT<T>::EPC;
#pragma unroll
  for (int c = 0; c < D / EPC; ++c) *reinterpret_cast<uint4*>(p + c * EPC) = f_to_chunk<T>(f + c * EPC);
}

// cooperative: rows [r0, r0 + nrows) of a strided [L, 32] matrix -> LDS floats [nrows][32]; rows >= L are zero
template <typename T, int NTHREADS>
__device__ __forceinline__ void load_tile(float* s, const T* base, int64_t rs, int r0, int nrows, int L, int tid) {
  constexpr int EPC = DT<T>::EPC, CPR = D / EPC;
  for (int c = tid; c < nrows * CPR; c += NTHREADS) {
    const int r = c / CPR, cc = c % CPR;
    float f[EPC];
    if (r0 + r < L) chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + (int64_t)(r0 + r) * rs + cc * EPC), f);
    else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) f[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < EPC; e += 4) *reinterpret_cast<float4*>(s + r * D + cc * EPC + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
  }
}

__device__ __forceinline__ float dot32(const float* a_reg, const float* s_row) {
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const float4 kv = *reinterpret_cast<const float4*>(s_row + d);
    acc += a_reg[d] * kv.x + a_reg[d + 1] * kv.y + a_reg[d + 2] * kv.z + a_reg[d + 3] * kv.w;
  }
  return acc;
}
__device__ __forceinline__ void axpy32(float* acc, float a, const float* s_row) {
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const float4 vv = *reinterpret_cast<const float4*>(s_row + d);
    acc[d] += a * vv.x; acc[d + 1] += a * vv.y; acc[d + 2] += a * vv.z; acc[d + 3] += a * vv.w;
  }
}

// =============================================================================================
// forward, one query per lane, keys streamed through LDS, online softmax over groups of 8 keys
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(QB) void fwd_manyq_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float sK[KC * D], sV[KC * D];
  const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * QB + tid;
  const bool valid = i < a.Lq;
  const T* qb = reinterpret_cast<const T*>(a.q) + b * a.q_bs + h * D;
  const T* kb = reinterpret_cast<const T*>(a.k) + b * a.k_bs + h * D;
  const T* vb = reinterpret_cast<const T*>(a.v) + b * a.v_bs + h * D;
  float q[D], o[D];
  if (valid) load_row<T>(qb + (int64_t)i * a.q_rs, q);
#pragma unroll
  for (int d = 0; d < D; ++d) { q[d] = valid ? q[d] * a.scale : 0.f; o[d] = 0.f; }
  float m = -1e30f, l = 0.f;
  for (int k0 = 0; k0 < a.Lk; k0 += KC) {
    __syncthreads();
    load_tile<T, QB>(sK, kb, a.k_rs, k0, KC, a.Lk, tid);
    load_tile<T, QB>(sV, vb, a.v_rs, k0, KC, a.Lk, tid);
    __syncthreads();
    const int nk = min(KC, a.Lk - k0);
    for (int j0 = 0; j0 < nk; j0 += 8) {
      float s[8], mx = -1e30f;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        s[jj] = (j0 + jj < nk) ? dot32(q, sK + (j0 + jj) * D) : -1e30f;
        mx = fmaxf(mx, s[jj]);
      }
      const float mn = fmaxf(m, mx), alpha = expf(m - mn);
      l *= alpha;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] *= alpha;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float p = (j0 + jj < nk) ? expf(s[jj] - mn) : 0.f;
        l += p;
        axpy32(o, p, sV + (j0 + jj) * D);
      }
      m = mn;
    }
  }
  if (valid) {
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] *= inv;
    store_row<T>(reinterpret_cast<T*>(a.o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D, o);
    if (a.lse) a.lse[((int64_t)b * a.H + h) * a.Lq + i] = m + logf(l);
  }
}

// =============================================================================================
// forward, <= 16 queries over many keys: one key per lane, keys split over workgroups,
// partial (m, l, o) per split + combine
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(FKB) void fwd_fewq_kernel(const Args a, float* __restrict__ part, int nsplit) {
  __shared__ __attribute__((aligned(16))) float sQ[FQ * D], sP[FQ * FKB], sV[FKB * D], sRed[2][4][FQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.y, b = blockIdx.z, split = blockIdx.x;
  const T* qb = reinterpret_cast<const T*>(a.q) + b * a.q_bs + h * D;
  const T* kb = reinterpret_cast<const T*>(a.k) + b * a.k_bs + h * D;
  const T* vb = reinterpret_cast<const T*>(a.v) + b * a.v_bs + h * D;
  load_tile<T, FKB>(sQ, qb, a.q_rs, 0, FQ, a.Lq, tid);
  const int j = split * FKB + tid;
  const bool valid = j < a.Lk;
  float kr[D], vr[D];
  if (valid) { load_row<T>(kb + (int64_t)j * a.k_rs, kr); load_row<T>(vb + (int64_t)j * a.v_rs, vr); }
#pragma unroll
  for (int d = 0; d < D; ++d) { if (!valid) { kr[d] = 0.f; vr[d] = 0.f; } }
#pragma unroll
  for (int d = 0; d < D; d += 4) *reinterpret_cast<float4*>(sV + tid * D + d) = make_float4(vr[d], vr[d + 1], vr[d + 2], vr[d + 3]);
  __syncthreads();
  float s[FQ];
#pragma unroll
  for (int i = 0; i < FQ; ++i) {
    s[i] = valid ? dot32(kr, sQ + i * D) * a.scale : -1e30f;
    const float wm = wave_max(s[i]);
    if (lane == 0) sRed[0][wave][i] = wm;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < FQ; ++i) {
    const float mi = fmaxf(fmaxf(sRed[0][0][i], sRed[0][1][i]), fmaxf(sRed[0][2][i], sRed[0][3][i]));
    const float p = valid ? expf(s[i] - mi) : 0.f;
    sP[i * FKB + tid] = p;
    const float ws = wave_sum(p);
    if (lane == 0) sRed[1][wave][i] = ws;
  }
  __syncthreads();
  // o[i][d], d = 2 * (tid & 15) .. +1, i = tid >> 4
  const int i = tid >> 4, d = (tid & 15) * 2;
  float o0 = 0.f, o1 = 0.f;
  const float* pr = sP + i * FKB;
#pragma unroll 8
  for (int jj = 0; jj < FKB; ++jj) {
    const float p = pr[jj];
    const float2 vv = *reinterpret_cast<const float2*>(sV + jj * D + d);
    o0 += p * vv.x; o1 += p * vv.y;
  }
  float* pp = part + ((((int64_t)b * a.H + h) * nsplit + split) * FQ + i) * (D + 2);
  pp[2 + d] = o0; pp[3 + d] = o1;
  if ((tid & 15) == 0) {
    pp[0] = fmaxf(fmaxf(sRed[0][0][i], sRed[0][1][i]), fmaxf(sRed[0][2][i], sRed[0][3][i]));
    pp[1] = sRed[1][0][i] + sRed[1][1][i] + sRed[1][2][i] + sRed[1][3][i];
  }
}

template <typename T>
__global__ __launch_bounds__(FQ * D) void fwd_fewq_combine_kernel(const Args a, const float* __restrict__ part, int nsplit) {
  const int i = threadIdx.x >> 5, d = threadIdx.x & 31, h = blockIdx.x, b = blockIdx.y;
  if (i >= a.Lq) return;
  const float* pp = part + (((int64_t)b * a.H + h) * nsplit * FQ + i) * (D + 2);
  const int64_t stride = (int64_t)FQ * (D + 2);
  float m = -1e30f;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, pp[s * stride]);
  float l = 0.f, o = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float w = expf(pp[s * stride] - m);
    l += pp[s * stride + 1] * w;
    o += pp[s * stride + 2 + d] * w;
  }
  DT<T>::st(reinterpret_cast<T*>(a.o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D + d, o / l);
  if (a.lse && d == 0) a.lse[((int64_t)b * a.H + h) * a.Lq + i] = m + logf(l);
}

// =============================================================================================
// backward
// =============================================================================================
// delta[b,h,i] = sum_d dO[b,i,h,d] * O[b,i,h,d]
template <typename T>
__global__ __launch_bounds__(256) void bwd_delta_kernel(const Args a, float* __restrict__ delta) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)a.B * a.H * a.Lq;
  if (idx >= total) return;
  const int i = (int)(idx % a.Lq), h = (int)((idx / a.Lq) % a.H), b = (int)(idx / ((int64_t)a.Lq * a.H));
  float o[D], g[D];
  load_row<T>(reinterpret_cast<const T*>(a.o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D, o);
  load_row<T>(reinterpret_cast<const T*>(a.d_o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D, g);
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) s += o[d] * g[d];
  delta[idx] = s;
}

// one key per lane: dK, dV (and, FEWQ, the <=16-row dQ through an LDS-staged reduction)
template <typename T, bool FEWQ>
__global__ __launch_bounds__(QB) void bwd_dkv_kernel(const Args a, const float* __restrict__ delta, float* __restrict__ dq_acc) {
  __shared__ __attribute__((aligned(16))) float sQ[KC * D], sG[KC * D], sL[KC], sDl[KC];
  __shared__ __attribute__((aligned(16))) float sDS[FEWQ ? FQ * QB : 1], sKk[FEWQ ? QB * D : 1];
  const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * QB + tid;
  const bool valid = j < a.Lk;
  const T* qb = reinterpret_cast<const T*>(a.q) + b * a.q_bs + h * D;
  const T* gb = reinterpret_cast<const T*>(a.d_o) + b * a.o_bs + h * D;
  float kr[D], vr[D], dk[D], dv[D];
  if (valid) {
    load_row<T>(reinterpret_cast<const T*>(a.k) + b * a.k_bs + (int64_t)j * a.k_rs + h * D, kr);
    load_row<T>(reinterpret_cast<const T*>(a.v) + b * a.v_bs + (int64_t)j * a.v_rs + h * D, vr);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) { if (!valid) { kr[d] = 0.f; vr[d] = 0.f; } dk[d] = 0.f; dv[d] = 0.f; }
  const int64_t bh = ((int64_t)b * a.H + h) * a.Lq;
  for (int q0 = 0; q0 < a.Lq; q0 += KC) {
    __syncthreads();
    load_tile<T, QB>(sQ, qb, a.q_rs, q0, KC, a.Lq, tid);
    load_tile<T, QB>(sG, gb, a.o_rs, q0, KC, a.Lq, tid);
    if (tid < KC) {
      const bool ok = q0 + tid < a.Lq;
      sL[tid] = ok ? a.lse[bh + q0 + tid] : 0.f;
      sDl[tid] = ok ? delta[bh + q0 + tid] : 0.f;
    }
    __syncthreads();
    const int nq = min(KC, a.Lq - q0);
    for (int ii = 0; ii < nq; ++ii) {
      const float s = dot32(kr, sQ + ii * D) * a.scale;
      const float p = valid ? expf(s - sL[ii]) : 0.f;
      axpy32(dv, p, sG + ii * D);
      const float dp = dot32(vr, sG + ii * D);
      const float ds = p * (dp - sDl[ii]) * a.scale;
      axpy32(dk, ds, sQ + ii * D);
      if (FEWQ) sDS[ii * QB + tid] = ds;
    }
  }
  if (valid) {
    store_row<T>(reinterpret_cast<T*>(a.dk) + b * a.k_bs + (int64_t)j * a.k_rs + h * D, dk);
    store_row<T>(reinterpret_cast<T*>(a.dv) + b * a.v_bs + (int64_t)j * a.v_rs + h * D, dv);
  }
  if (FEWQ) {
    // dQ[i][d] += sum_j ds[i][j] * K[j][d] over this workgroup's keys, then one atomic per output
#pragma unroll
    for (int d = 0; d < D; d += 4) *reinterpret_cast<float4*>(sKk + tid * D + d) = make_float4(kr[d], kr[d + 1], kr[d + 2], kr[d + 3]);
    __syncthreads();
    const int i = tid >> 3, d0 = (tid & 7) * 4;
    if (i < a.Lq) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      const float* dsr = sDS + i * QB;
#pragma unroll 8
      for (int jj = 0; jj < QB; ++jj) {
        const float w = dsr[jj];
        const float4 kk = *reinterpret_cast<const float4*>(sKk + jj * D + d0);
        acc[0] += w * kk.x; acc[1] += w * kk.y; acc[2] += w * kk.z; acc[3] += w * kk.w;
      }
      float* dst = dq_acc + (((int64_t)b * a.H + h) * FQ + i) * D + d0;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(dst + e, acc[e]);
    }
  }
}

// one query per lane: dQ (and, FEWK, the <=16-row dK / dV through an LDS-staged reduction)
template <typename T, bool FEWK>
__global__ __launch_bounds__(QB) void bwd_dq_kernel(const Args a, const float* __restrict__ delta, float* __restrict__ dk_acc,
                                                   float* __restrict__ dv_acc) {
  __shared__ __attribute__((aligned(16))) float sK[KC * D], sV[KC * D];
  __shared__ __attribute__((aligned(16))) float sP[FEWK ? QB * FQ : 1], sDS[FEWK ? QB * FQ : 1], sQq[FEWK ? QB * D : 1], sGg[FEWK ? QB * D : 1];
  const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * QB + tid;
  const bool valid = i < a.Lq;
  const T* kb = reinterpret_cast<const T*>(a.k) + b * a.k_bs + h * D;
  const T* vb = reinterpret_cast<const T*>(a.v) + b * a.v_bs + h * D;
  float q[D], g[D], dq[D];
  float lse = 0.f, dl = 0.f;
  if (valid) {
    load_row<T>(reinterpret_cast<const T*>(a.q) + b * a.q_bs + (int64_t)i * a.q_rs + h * D, q);
    load_row<T>(reinterpret_cast<const T*>(a.d_o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D, g);
    lse = a.lse[((int64_t)b * a.H + h) * a.Lq + i];
    dl = delta[((int64_t)b * a.H + h) * a.Lq + i];
  }
#pragma unroll
  for (int d = 0; d < D; ++d) { if (!valid) { q[d] = 0.f; g[d] = 0.f; } dq[d] = 0.f; }
  for (int k0 = 0; k0 < a.Lk; k0 += KC) {
    __syncthreads();
    load_tile<T, QB>(sK, kb, a.k_rs, k0, KC, a.Lk, tid);
    load_tile<T, QB>(sV, vb, a.v_rs, k0, KC, a.Lk, tid);
    __syncthreads();
    const int nk = min(KC, a.Lk - k0);
    for (int jj = 0; jj < nk; ++jj) {
      const float s = dot32(q, sK + jj * D) * a.scale;
      const float p = valid ? expf(s - lse) : 0.f;
      const float dp = dot32(g, sV + jj * D);
      const float ds = p * (dp - dl) * a.scale;
      axpy32(dq, ds, sK + jj * D);
      if (FEWK) { sP[tid * FQ + jj] = p; sDS[tid * FQ + jj] = ds; }
    }
  }
  if (valid) store_row<T>(reinterpret_cast<T*>(a.dq) + b * a.q_bs + (int64_t)i * a.q_rs + h * D, dq);
  if (FEWK) {
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      *reinterpret_cast<float4*>(sQq + tid * D + d) = make_float4(q[d], q[d + 1], q[d + 2], q[d + 3]);
      *reinterpret_cast<float4*>(sGg + tid * D + d) = make_float4(g[d], g[d + 1], g[d + 2], g[d + 3]);
    }
    __syncthreads();
    const int j = tid >> 3, d0 = (tid & 7) * 4;
    if (j < a.Lk) {
      float av[4] = {0.f, 0.f, 0.f, 0.f}, ak[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int ii = 0; ii < QB; ++ii) {
        const float p = sP[ii * FQ + j], ds = sDS[ii * FQ + j];
        const float4 gg = *reinterpret_cast<const float4*>(sGg + ii * D + d0);
        const float4 qq = *reinterpret_cast<const float4*>(sQq + ii * D + d0);
        av[0] += p * gg.x; av[1] += p * gg.y; av[2] += p * gg.z; av[3] += p * gg.w;
        ak[0] += ds * qq.x; ak[1] += ds * qq.y; ak[2] += ds * qq.z; ak[3] += ds * qq.w;
      }
      const int64_t o = (((int64_t)b * a.H + h) * FQ + j) * D + d0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { atomicAdd(dv_acc + o + e, av[e]); atomicAdd(dk_acc + o + e, ak[e]); }
    }
  }
}

// fp32 [B][H][FQ][32] accumulator -> strided T rows
template <typename T>
__global__ __launch_bounds__(256) void scatter_kernel(const float* __restrict__ acc, T* __restrict__ dst, int64_t bs, int64_t rs, int B, int H, int L) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * H * L * D;
  if (idx >= total) return;
  const int d = (int)(idx % D), l = (int)((idx / D) % L), h = (int)((idx / ((int64_t)D * L)) % H), b = (int)(idx / ((int64_t)D * L * H));
  DT<T>::st(dst + b * bs + (int64_t)l * rs + h * D + d, acc[(((int64_t)b * H + h) * FQ + l) * D + d]);
}

inline bool few_q(int Lq, int Lk) { return Lq <= FQ && Lk > KC; }
inline bool few_k(int Lq, int Lk) { return Lk <= FQ && Lq > KC; }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int check_desc(const lmv_attn_desc* d, int dtype, bool bwd) {
  if (!d) LMV_FAIL(LMV_ERR_SHAPE, "attn: null descriptor");
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "attn: unsupported dtype %d", dtype);
  if (d->B <= 0 || d->H <= 0 || d->Lq <= 0 || d->Lk <= 0) LMV_FAIL(LMV_ERR_SHAPE, "attn: bad sizes B=%d H=%d Lq=%d Lk=%d", d->B, d->H, d->Lq, d->Lk);
  if (d->H > 65535 || d->B > 65535) LMV_FAIL(LMV_ERR_SHAPE, "attn: B and H must be <= 65535");
  if (!(d->scale > 0.f)) LMV_FAIL(LMV_ERR_SHAPE, "attn: scale must be positive (the softmax maximum is taken on the raw scores)");
  const int64_t st[8] = {d->q_bs, d->q_rs, d->k_bs, d->k_rs, d->v_bs, d->v_rs, d->o_bs, d->o_rs};
  for (int i = 0; i < 8; ++i)
    if (st[i] % 8) LMV_FAIL(LMV_ERR_SHAPE, "attn: strides must be multiples of 8 elements");
  if (!d->q || !d->k || !d->v || !d->o || !lmv_aligned16(d->q) || !lmv_aligned16(d->k) || !lmv_aligned16(d->v) || !lmv_aligned16(d->o))
    LMV_FAIL(LMV_ERR_SHAPE, "attn: null or misaligned operand");
  if (bwd && (!d->lse || !d->d_o || !d->dq || !d->dk || !d->dv || !lmv_aligned16(d->d_o) || !lmv_aligned16(d->dq) || !lmv_aligned16(d->dk) || !lmv_aligned16(d->dv)))
    LMV_FAIL(LMV_ERR_SHAPE, "attn_bwd: null or misaligned gradient operand");
  return LMV_OK;
}

Args to_args(const lmv_attn_desc* d) {
  Args a;
  a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o; a.lse = d->lse; a.d_o = d->d_o; a.dq = d->dq; a.dk = d->dk; a.dv = d->dv;
  a.q_bs = d->q_bs; a.q_rs = d->q_rs; a.k_bs = d->k_bs; a.k_rs = d->k_rs; a.v_bs = d->v_bs; a.v_rs = d->v_rs; a.o_bs = d->o_bs; a.o_rs = d->o_rs;
  a.B = d->B; a.H = d->H; a.Lq = d->Lq; a.Lk = d->Lk; a.scale = d->scale;
  return a;
}

template <typename T>
int fwd_impl(const lmv_attn_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
  const Args a = to_args(d);
  if (sizeof(T) == 2 && lmv_attn_mfma_supported(a)) return lmv_attn_mfma_fwd(a, st);
  if (sizeof(T) == 2 && lmv_attn_mfma_long_supported(a)) return lmv_attn_mfma_long_fwd(a, st);
  if (sizeof(T) == 2 && lmv_attn_mfma_fewq_supported(a)) {
    const int nsplit = lmv_attn_mfma_fewq_nsplit(a);
    const size_t need = (size_t)d->B * d->H * nsplit * FQ * (D + 2) * sizeof(float);
    if (!ws || ws_bytes < need) LMV_FAIL(LMV_ERR_WORKSPACE, "attn_fwd: workspace %zu < %zu bytes", ws_bytes, need);
    if (int rc = lmv_attn_mfma_fewq_fwd(a, (float*)ws, st)) return rc;
    hipLaunchKernelGGL((fwd_fewq_combine_kernel<T>), dim3(d->H, d->B), dim3(FQ * D), 0, st, a, (const float*)ws, nsplit);
    LMV_CHECK_LAUNCH("attn_fwd combine");
    return LMV_OK;
  }
  if (few_q(d->Lq, d->Lk)) {
    const int nsplit = (d->Lk + FKB - 1) / FKB;
    const size_t need = (size_t)d->B * d->H * nsplit * FQ * (D + 2) * sizeof(float);
    if (!ws || ws_bytes < need) LMV_FAIL(LMV_ERR_WORKSPACE, "attn_fwd: workspace %zu < %zu bytes", ws_bytes, need);
    hipLaunchKernelGGL((fwd_fewq_kernel<T>), dim3(nsplit, d->H, d->B), dim3(FKB), 0, st, a, (float*)ws, nsplit);
    hipLaunchKernelGGL((fwd_fewq_combine_kernel<T>), dim3(d->H, d->B), dim3(FQ * D), 0, st, a, (const float*)ws, nsplit);
  } else {
    hipLaunchKernelGGL((fwd_manyq_kernel<T>), dim3((d->Lq + QB - 1) / QB, d->H, d->B), dim3(QB), 0, st, a);
  }
  LMV_CHECK_LAUNCH("attn_fwd");
  return LMV_OK;
}

template <typename T>
int bwd_impl(const lmv_attn_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
  const Args a = to_args(d);
  const size_t need = lmv_attn_workspace_bytes(d->B, d->H, d->Lq, d->Lk, 1);
  if (!ws || ws_bytes < need) LMV_FAIL(LMV_ERR_WORKSPACE, "attn_bwd: workspace %zu < %zu bytes", ws_bytes, need);
  float* delta = (float*)ws;
  const size_t delta_bytes = align256((size_t)d->B * d->H * d->Lq * sizeof(float));
  const size_t acc_elems = (size_t)d->B * d->H * FQ * D;
  float* acc0 = (float*)((char*)ws + delta_bytes);
  float* acc1 = acc0 + acc_elems;
  const int64_t nd = (int64_t)d->B * d->H * d->Lq;
  // the bf16 MFMA paths compute delta = rowsum(dO * O) inside their own kernels
  if (sizeof(T) == 2 && lmv_attn_mfma_supported(a)) return lmv_attn_mfma_bwd(a, delta, acc0, st);
  if (sizeof(T) == 2 && lmv_attn_mfma_long_supported(a)) return lmv_attn_mfma_long_bwd(a, delta, st);
  if (sizeof(T) == 2 && lmv_attn_mfma_fewq_supported(a)) return lmv_attn_mfma_fewq_bwd(a, acc0, st);
  hipLaunchKernelGGL((bwd_delta_kernel<T>), dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, a, delta);
  const dim3 gk((d->Lk + QB - 1) / QB, d->H, d->B), gq((d->Lq + QB - 1) / QB, d->H, d->B);
  if (few_q(d->Lq, d->Lk)) {
    if (hipMemsetAsync(acc0, 0, acc_elems * sizeof(float), st) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "attn_bwd: memset failed");
    hipLaunchKernelGGL((bwd_dkv_kernel<T, true>), gk, dim3(QB), 0, st, a, (const float*)delta, acc0);
    const int64_t n = (int64_t)d->B * d->H * d->Lq * D;
    hipLaunchKernelGGL((scatter_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)acc0, (T*)d->dq, d->q_bs, d->q_rs, d->B, d->H, d->Lq);
  } else if (few_k(d->Lq, d->Lk)) {
    if (hipMemsetAsync(acc0, 0, 2 * acc_elems * sizeof(float), st) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "attn_bwd: memset failed");
    hipLaunchKernelGGL((bwd_dq_kernel<T, true>), gq, dim3(QB), 0, st, a, (const float*)delta, acc0, acc1);
    const int64_t n = (int64_t)d->B * d->H * d->Lk * D;
    hipLaunchKernelGGL((scatter_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)acc0, (T*)d->dk, d->k_bs, d->k_rs, d->B, d->H, d->Lk);
    hipLaunchKernelGGL((scatter_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)acc1, (T*)d->dv, d->v_bs, d->v_rs, d->B, d->H, d->Lk);
  } else {
    hipLaunchKernelGGL((bwd_dkv_kernel<T, false>), gk, dim3(QB), 0, st, a, (const float*)delta, (float*)nullptr);
    hipLaunchKernelGGL((bwd_dq_kernel<T, false>), gq, dim3(QB), 0, st, a, (const float*)delta, (float*)nullptr, (float*)nullptr);
  }
  LMV_CHECK_LAUNCH("attn_bwd");
  return LMV_OK;
}

}  // namespace

extern "C" size_t lmv_attn_workspace_bytes(int B, int H, int Lq, int Lk, int backward) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  if (!backward) {
    if (!few_q(Lq, Lk)) return 256;
    const int nsplit = (Lk + 127) / 128;        // the bf16 MFMA path splits the keys into ranges of 128 (generic path: 256)
    return align256((size_t)B * H * nsplit * FQ * (D + 2) * sizeof(float));
  }
  // delta + fp32 accumulators of the split reductions (sized for the larger of the generic and the MFMA path)
  AttnArgs a{};
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  size_t acc = 2 * (size_t)B * H * FQ * D * sizeof(float);
  if (lmv_attn_mfma_supported(a) && lmv_attn_mfma_bwd_acc_bytes(a) > acc) acc = lmv_attn_mfma_bwd_acc_bytes(a);
  if (few_q(Lq, Lk) && lmv_attn_mfma_fewq_bwd_acc_bytes(a) > acc) acc = lmv_attn_mfma_fewq_bwd_acc_bytes(a);     // dQ slabs of the split-key path
  return align256((size_t)B * H * Lq * sizeof(float)) + align256(acc);
}

// (timing probe) FLOPs / algorithmic bytes of one attention problem: forward 2 contractions, backward 5 (S and dP recomputed, dV, dK, dQ); bytes: q, k, v, o (+ dO, dq, dk, dv)
static inline double attn_flops(const lmv_attn_desc* d, bool bwd) { return (bwd ? 10.0 : 4.0) * d->B * d->H * (double)d->Lq * d->Lk * 32.0; }
static inline double attn_bytes(const lmv_attn_desc* d, int dtype, bool bwd) {
  const double es = dtype == LMV_BF16 ? 2.0 : 4.0, rows = (double)d->B * d->H * 32.0 * (2.0 * d->Lq + 2.0 * d->Lk);
  return rows * es * (bwd ? 2.0 : 1.0) + (bwd ? (double)d->B * d->H * d->Lq * 32.0 * es : 0.0);
}
/* Two independent attention problems (same B and H): the image-token and the meta-token self-attention of an S block.  bf16 problems
 * of the model's hot shapes run as ONE launch; anything else as two.  The workspace must satisfy lmv_attn_workspace_bytes of both. */
extern "C" int lmv_attn_fwd_pair(const lmv_attn_desc* d, void* ws, size_t ws_bytes, int dtype, void* stream) {
  for (int i = 0; i < 2; ++i)
    if (int rc = check_desc(d + i, dtype, false)) return rc;
  LmvTimedLaunch timed(stream, attn_flops(d, false) + attn_flops(d + 1, false), attn_bytes(d, dtype, false) + attn_bytes(d + 1, dtype, false), LMV_TK_ATTN_FWD);
  if (dtype == LMV_BF16) {
    const Args a1 = to_args(d), a2 = to_args(d + 1);
    if (lmv_attn_mfma_supported(a1) && lmv_attn_mfma_supported(a2) && lmv_attn_mfma_fwd_pair(a1, a2, (hipStream_t)stream)) {
      LMV_CHECK_LAUNCH("attn_fwd_pair");
      return LMV_OK;
    }
  }
  for (int i = 0; i < 2; ++i) {
    const int rc = dtype == LMV_BF16 ? fwd_impl<bf16_t>(d + i, ws, ws_bytes, (hipStream_t)stream) : fwd_impl<float>(d + i, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
  }
  return LMV_OK;
}
extern "C" int lmv_attn_bwd_pair(const lmv_attn_desc* d, void* ws, size_t ws_bytes, int dtype, void* stream) {
  for (int i = 0; i < 2; ++i)
    if (int rc = check_desc(d + i, dtype, true)) return rc;
  LmvTimedLaunch timed(stream, attn_flops(d, true) + attn_flops(d + 1, true), attn_bytes(d, dtype, true) + attn_bytes(d + 1, dtype, true), LMV_TK_ATTN_BWD);
  if (dtype == LMV_BF16) {
    const Args a1 = to_args(d), a2 = to_args(d + 1);
    if (lmv_attn_mfma_supported(a1) && lmv_attn_mfma_supported(a2) && lmv_attn_mfma_bwd_pair(a1, a2, (hipStream_t)stream)) {
      LMV_CHECK_LAUNCH("attn_bwd_pair");
      return LMV_OK;
    }
  }
  for (int i = 0; i < 2; ++i) {
    const int rc = dtype == LMV_BF16 ? bwd_impl<bf16_t>(d + i, ws, ws_bytes, (hipStream_t)stream) : bwd_impl<float>(d + i, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
  }
  return LMV_OK;
}

extern "C" int lmv_attn_fwd(const lmv_attn_desc* d, void* ws, size_t ws_bytes, int dtype, void* stream) {
  if (int rc = check_desc(d, dtype, false)) return rc;
  LmvTimedLaunch timed(stream, attn_flops(d, false), attn_bytes(d, dtype, false), LMV_TK_ATTN_FWD);
  return dtype == LMV_BF16 ? fwd_impl<bf16_t>(d, ws, ws_bytes, (hipStream_t)stream) : fwd_impl<float>(d, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int lmv_attn_bwd(const lmv_attn_desc* d, void* ws, size_t ws_bytes, int dtype, void* stream) {
  if (int rc = check_desc(d, dtype, true)) return rc;
  LmvTimedLaunch timed(stream, attn_flops(d, true), attn_bytes(d, dtype, true), LMV_TK_ATTN_BWD);
  return dtype == LMV_BF16 ? bwd_impl<bf16_t>(d, ws, ws_bytes, (hipStream_t)stream) : bwd_impl<float>(d, ws, ws_bytes, (hipStream_t)stream);
}

// ---- named cores of the reference seam ------------------------------------------------------
static int head_count(int C, const char* who) {
  if (C <= 0 || (C % D)) { lmv_set_error("%s: C=%d must be a positive multiple of the head dim 32", who, C); return -1; }
  return C / D;
}

extern "C" int lmv_sa_core_fwd(const void* qkv, void* o, float* lse, int B, int L, int C, void* ws, size_t ws_bytes, int dtype, void* stream) {
  const int H = head_count(C, "sa_core"); if (H < 0) return LMV_ERR_SHAPE;
  const size_t es = dtype == LMV_BF16 ? 2 : 4;
  lmv_attn_desc d{};
  d.q = qkv; d.k = (const char*)qkv + (size_t)C * es; d.v = (const char*)qkv + 2 * (size_t)C * es; d.o = o; d.lse = lse;
  d.q_bs = d.k_bs = d.v_bs = (int64_t)L * 3 * C; d.q_rs = d.k_rs = d.v_rs = 3 * C; d.o_bs = (int64_t)L * C; d.o_rs = C;
  d.B = B; d.H = H; d.Lq = L; d.Lk = L; d.scale = 1.f / sqrtf((float)D);
  return lmv_attn_fwd(&d, ws, ws_bytes, dtype, stream);
}

extern "C" int lmv_ca_core_fwd(const void* q, const void* kv, void* o, float* lse, int B, int M, int N, int C, void* ws, size_t ws_bytes,
                               int dtype, void* stream) {
  const int H = head_count(C, "ca_core"); if (H < 0) return LMV_ERR_SHAPE;
  const size_t es = dtype == LMV_BF16 ? 2 : 4;
  lmv_attn_desc d{};
  d.q = q; d.k = kv; d.v = (const char*)kv + (size_t)C * es; d.o = o; d.lse = lse;
  d.q_bs = (int64_t)M * C; d.q_rs = C; d.k_bs = d.v_bs = (int64_t)N * 2 * C; d.k_rs = d.v_rs = 2 * C; d.o_bs = (int64_t)M * C; d.o_rs = C;
  d.B = B; d.H = H; d.Lq = M; d.Lk = N; d.scale = 1.f / sqrtf((float)D);
  return lmv_attn_fwd(&d, ws, ws_bytes, dtype, stream);
}

extern "C" int lmv_dca_core_fwd(const void* qkv1, const void* qkv2, void* ox, void* oc, float* lse_x, float* lse_c, int B, int N, int M, int C,
                                void* ws, size_t ws_bytes, int dtype, void* stream) {
  const int H = head_count(C, "dca_core"); if (H < 0) return LMV_ERR_SHAPE;
  if (N < 2) LMV_FAIL(LMV_ERR_SHAPE, "dca_core: N must be >= 2 (scale_x = log_N(M))");
  const size_t es = dtype == LMV_BF16 ? 2 : 4;
  const float base = 1.f / sqrtf((float)C);   // models/lemevit.py:235 -- the FULL embed dim
  lmv_attn_desc d{};
  // image tokens attend to meta tokens: q1, k2, v2
  d.q = qkv1; d.k = (const char*)qkv2 + (size_t)C * es; d.v = (const char*)qkv2 + 2 * (size_t)C * es; d.o = ox; d.lse = lse_x;
  d.q_bs = (int64_t)N * 3 * C; d.q_rs = 3 * C; d.k_bs = d.v_bs = (int64_t)M * 3 * C; d.k_rs = d.v_rs = 3 * C; d.o_bs = (int64_t)N * C; d.o_rs = C;
  d.B = B; d.H = H; d.Lq = N; d.Lk = M; d.scale = (float)(log((double)M) / log((double)N)) * base;
  if (int rc = lmv_attn_fwd(&d, ws, ws_bytes, dtype, stream)) return rc;
  // meta tokens attend to image tokens: q2, k1, v1
  d.q = qkv2; d.k = (const char*)qkv1 + (size_t)C * es; d.v = (const char*)qkv1 + 2 * (size_t)C * es; d.o = oc; d.lse = lse_c;
  d.q_bs = (int64_t)M * 3 * C; d.k_bs = d.v_bs = (int64_t)N * 3 * C; d.o_bs = (int64_t)M * C;
  d.Lq = M; d.Lk = N; d.scale = base;
  return lmv_attn_fwd(&d, ws, ws_bytes, dtype, stream);
}
