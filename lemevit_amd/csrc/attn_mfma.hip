// attn_mfma.hip -- bf16 MFMA attention cores (head dim 32) for key counts up to 224:
// the late-stage self-attention (L = 196 / 49 / 16) and the image->meta direction of Dual Cross-Attention
// (N queries x 16 keys).  v_mfma_f32_16x16x32_bf16 fits d = 32 exactly: one instruction per 16x16 score tile.
//
// Register-level dataflow (no LDS round trip for the probabilities):
//   S^T tile  D[key][q] = mfma(A = K fragment, B = Q fragment)       lane holds 4 consecutive keys of query (lane & 15)
//   softmax over the whole row in registers: 4 * NKT values per lane + two xor-shuffles across the 4 lane groups
//   O^T tile  D[d][q]   = sum over 32-key blocks of mfma(A = V^T fragment, B = P fragment)
// The P fragment of a 32-key block is exactly the two S^T accumulators of that block converted to bf16 -- the MFMA
// contraction does not care WHICH keys sit in which k-slot as long as both operands agree -- and the matching V^T
// fragment is what two ds_read_b64_tr_b16 transpose reads deliver (4 keys x 16 d per 16-lane group).
// The backward kernels use the same trick in both orientations (D[key][q] for dQ, D[q][key] for dK / dV) and
// recompute P = exp(scale * s - LSE).
//
// LDS images are [rows][32] bf16 (64-byte rows); the 16-byte chunk c of row r is stored at c ^ (((r >> 2) & 1) << 1),
// which is conflict-free for BOTH the 16-byte fragment reads and the transpose reads (derivation in DESIGN.md).
#include <stdlib.h>
#include <atomic>
#include "common.h"
#include "attn_internal.h"

#ifndef LMV_XCD_REMAP
#define LMV_XCD_REMAP 1
#endif

namespace {

constexpr int D = 32;
typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;

__device__ __forceinline__ int swz(int r, int c16) { return r * 64 + ((c16 ^ (((r >> 2) & 1) << 1)) << 4); }

// rows [r0, r0 + nrows) of TWO strided [L][32] bf16 matrices (K and V, or Q and dO) -> their LDS images (rows >= L zero-filled).
// Every thread fetches a batch of 4 chunks of each matrix from CLAMPED row indices with no branch around the loads, so the 8
// loads of a batch are in flight together (a bounds branch per chunk costs one L2 / HBM round trip per chunk: the staging of a
// 224-row K / V pair was 7 serialised round trips); out-of-range rows are zeroed on the way into LDS.
template <int NTHREADS>
__device__ __forceinline__ void stage_rows2(unsigned char* sa, const bf16_t* ba, int64_t rsa, unsigned char* sb, const bf16_t* bb, int64_t rsb,
                                            int r0, int nrows, int L, int tid) {
  const int total = nrows * 4, last = max(L - 1, 0);
  for (int c0 = 0; c0 < total; c0 += NTHREADS * 4) {
    uint4 va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = min(c0 + j * NTHREADS + tid, total - 1), cc = c & 3;
      const int rr = min(r0 + (c >> 2), last);
      va[j] = *reinterpret_cast<const uint4*>(ba + (int64_t)rr * rsa + cc * 8);
      vb[j] = *reinterpret_cast<const uint4*>(bb + (int64_t)rr * rsb + cc * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + j * NTHREADS + tid, r = c >> 2, cc = c & 3;
      if (c >= total) continue;
      const bool ok = r0 + r < L;
      *reinterpret_cast<uint4*>(sa + swz(r, cc)) = ok ? va[j] : make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(sb + swz(r, cc)) = ok ? vb[j] : make_uint4(0, 0, 0, 0);
    }
  }
}

// normal fragment: 16 rows starting at row0, lane -> row (lane & 15), 8 d-values at (lane >> 4) * 8
__device__ __forceinline__ bf16x8_t frag_n(const unsigned char* s, int row0, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(s + swz(row0 + (lane & 15), lane >> 4));
}
// transposed fragment for the 32-row block {rowA + 0..15, rowB + 0..15} and one HALF (d0 = 0 / 16 selects half 0 / 1) of the 32
// columns, in the INTERLEAVED order  lane i -> column 8 * (i >> 2) + 4 * half + (i & 3):  half 0 holds d = {0-3, 8-11, 16-19,
// 24-27}, half 1 the rest.  An output tile pair (x0, x1) = (MFMA(frag_t(.., 0), ..), MFMA(frag_t(.., 16), ..)) then leaves lane
// group g with d = 8g .. 8g+3 in x0 and 8g+4 .. 8g+7 in x1 -- EIGHT consecutive d of one row = one 16-byte store (store8), and the
// four lane groups cover the row's 64 bytes.  (With the natural order a lane owned two 8-byte pieces 32 bytes apart: 8-byte stores
// in 32-byte segments ran at ~2 TB/s, tools/native/panel_probe.hip.)  The contraction only needs both operands to agree.
// k-slot j of lane group g -> row (j < 4 ? rowA : rowB) + g * 4 + (j & 3)
__device__ __forceinline__ bf16x8_t frag_t(const unsigned char* s, int rowA, int rowB, int d0, int lane) {
  const int g = lane >> 4, i = lane & 15, rr = i >> 2, q = i & 3;
  const int ra = rowA + g * 4 + rr, rb = rowB + g * 4 + rr, c16 = q, off = (d0 >> 4) * 8;
  bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(s + swz(ra, c16) + off));
  bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(s + swz(rb, c16) + off));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8_t pack8(const f32x4_t& a, const f32x4_t& b) {
  bf16x8_t r;
  r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
  r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
  return r;
}
__device__ __forceinline__ bf16x8_t load_frag_global(const bf16_t* base, int64_t rs, int row, int L, int lane) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < L) v = *reinterpret_cast<const uint4*>(base + (int64_t)row * rs + (lane >> 4) * 8);
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ float group_max4(float v) { return lmv_xmax4(v); }
__device__ __forceinline__ float group_sum4(float v) { return lmv_xsum4(v); }
__device__ __forceinline__ void store4(bf16_t* p, const f32x4_t& v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])); }
// the tile pair (x0, x1) of a frag_t-produced output: d = 8g .. 8g+7 of the row at `row` (32 d, 64 bytes)
__device__ __forceinline__ void store8(bf16_t* row, int g, const f32x4_t& x0, const f32x4_t& x1) {
  *reinterpret_cast<uint4*>(row + g * 8) = make_uint4(pack_bf2(x0[0], x0[1]), pack_bf2(x0[2], x0[3]), pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3]));
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// XCD-contiguous workgroup order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2).
// The heads of one image read 64-byte pieces of the SAME rows (two heads per 128-byte line) and write 64-byte pieces of the same
// output rows: dealt to different XCDs, every line is fetched from HBM by two L2s and written back in halves.  Remapped, XCD x works
// on the contiguous range [x * n / 8, (x + 1) * n / 8) of logical ids = whole images.  (Speed only; any placement is correct.)
__device__ __forceinline__ int xcd_contiguous(int id, int n) {
  if (n & 7) return id;
  return (id & 7) * (n >> 3) + (id >> 3);
}

// =============================================================================================
// forward: NKT = number of 16-key tiles (even), keys padded with zero rows / masked scores
//   LK  > 0: the key count is a compile-time constant (the model's hot shapes: 196 / 49 / 16 keys): tiles past the end are not
//            computed at all, the boundary tile is masked with one select per score, and nothing else is -- the run-time version
//            carries a compare / select pair per score and tile plus wave-uniform branches, ~1000 VALU / SALU instructions per
//            16 queries against 42 MFMAs (the kernel is VALU-issue bound).
//   PV16:    P and V enter the P V product as fp16 (v_mfma_f32_16x16x32_f16): P in [0, 1] keeps 11 mantissa bits, so the
//            hi + lo bf16 split of P (2 extra MFMAs and ~3 VALU per score) is not needed for the 1e-3 budget; V is converted once
//            per workgroup inside LDS (bf16 -> fp16 is exact; saturated at the fp16 range).
// =============================================================================================
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

__device__ __forceinline__ unsigned bf2_to_h2(unsigned w) {          // two bf16 -> two fp16 (exact inside the fp16 range, saturating)
  const float lo = fminf(fmaxf(__uint_as_float(w << 16), -65504.f), 65504.f), hi = fminf(fmaxf(__uint_as_float(w & 0xffff0000u), -65504.f), 65504.f);
  return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
}
__device__ __forceinline__ f16x8_t pack8h(const f32x4_t& a, const f32x4_t& b) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t w = {__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a[0], a[1])), __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a[2], a[3])),
                     __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(b[0], b[1])), __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(b[2], b[3]))};
  return __builtin_bit_cast(f16x8_t, w);
}

template <int NKT, int LK = 0, bool PV16 = false>
__device__ __forceinline__ void mfma_fwd_body(const AttnArgs& a, int qt_per_block, const int bx, const int h, const int b) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[NKT * 16 * 64], sV[NKT * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Lk = LK ? LK : a.Lk;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  bf16_t* ob = reinterpret_cast<bf16_t*>(a.o) + b * a.o_bs + h * D;
  stage_rows2<256>(sK, kb, a.k_rs, sV, vb, a.v_rs, 0, NKT * 16, Lk, tid);
  __syncthreads();
  if constexpr (PV16) {
    for (int c = tid; c < NKT * 16 * 4; c += 256) {
      uint4 v = *reinterpret_cast<const uint4*>(sV + c * 16);
      v.x = bf2_to_h2(v.x); v.y = bf2_to_h2(v.y); v.z = bf2_to_h2(v.z); v.w = bf2_to_h2(v.w);
      *reinterpret_cast<uint4*>(sV + c * 16) = v;
    }
    __syncthreads();
  }
  const int nqt = (a.Lq + 15) >> 4;
  const int qt_end = min(nqt, (bx + 1) * qt_per_block);
  const int g = lane >> 4;
  const float sc2 = a.scale * 1.4426950408889634f;
  // the query fragment of the NEXT iteration is fetched while this one computes
  bf16x8_t qf_next = load_frag_global(qb, a.q_rs, (bx * qt_per_block + wave) * 16 + (lane & 15), a.Lq, lane);
  for (int qt = bx * qt_per_block + wave; qt < qt_end; qt += 4) {
    const int q = qt * 16 + (lane & 15);
    const bf16x8_t qf = qf_next;
    if (qt + 4 < qt_end) qf_next = load_frag_global(qb, a.q_rs, q + 64, a.Lq, lane);
    // softmax on the RAW scores (scale > 0 commutes with the maximum): p = 2^(s * c - m * c), c = scale * log2(e) -- one FMA
    // and one v_exp_f32 per score
    f32x4_t s[NKT];
    float m = -3e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (LK && kt * 16 >= LK) { s[kt] = f32x4_t{-3e38f, -3e38f, -3e38f, -3e38f}; continue; }      // compile-time: padding tile
      s[kt] = MFMA(frag_n(sK, kt * 16, lane), qf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      if (kt * 16 + 16 > Lk) {                              // boundary / padding tile (wave-uniform; compile-time for LK > 0)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 16 + g * 4 + r >= Lk) s[kt][r] = -3e38f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
    }
    m = group_max4(m);
    const float mc = m * sc2;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (LK && kt * 16 >= LK) { s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; continue; }
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sc2, -mc)); s[kt][r] = p; l += p; }
    }
    l = group_sum4(l);
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb2 = 0; kb2 < NKT / 2; ++kb2) {
      if (LK && kb2 * 32 >= LK) continue;
      const bf16x8_t vt0 = frag_t(sV, kb2 * 32, kb2 * 32 + 16, 0, lane), vt1 = frag_t(sV, kb2 * 32, kb2 * 32 + 16, 16, lane);
      if constexpr (PV16) {
        const f16x8_t ph = pack8h(s[2 * kb2], s[2 * kb2 + 1]);
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, vt0), ph, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, vt1), ph, o1, 0, 0, 0);
      } else {
        // P is fed to the matrix core as hi + lo bf16 parts (16 mantissa bits): the forward output then carries only
        // its own bf16 rounding, which keeps the kernel inside the 1e-3 parity budget at 2 extra MFMAs per block
        const bf16x8_t ph = pack8(s[2 * kb2], s[2 * kb2 + 1]);
        f32x4_t r0, r1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { r0[r] = s[2 * kb2][r] - (float)ph[r]; r1[r] = s[2 * kb2 + 1][r] - (float)ph[4 + r]; }
        const bf16x8_t pl = pack8(r0, r1);
        o0 = MFMA(vt0, ph, o0); o0 = MFMA(vt0, pl, o0);
        o1 = MFMA(vt1, ph, o1); o1 = MFMA(vt1, pl, o1);
      }
    }
    if (q < a.Lq) {
      const float inv = 1.f / l;
      o0 *= inv; o1 *= inv;
      store8(ob + (int64_t)q * a.o_rs, g, o0, o1);
      if (a.lse && g == 0) a.lse[((int64_t)b * a.H + h) * a.Lq + q] = m * a.scale + __logf(l);
    }
  }
}

template <int NKT, int LK = 0, bool PV16 = false>
__global__ __launch_bounds__(256) void mfma_fwd_kernel(const AttnArgs a, int qt_per_block) {
  const int nb = gridDim.x, n = nb * a.H * a.B;
  const int L = LMV_XCD_REMAP ? xcd_contiguous((int)blockIdx.x + nb * ((int)blockIdx.y + a.H * (int)blockIdx.z), n) : (int)blockIdx.x + nb * ((int)blockIdx.y + a.H * (int)blockIdx.z);
  mfma_fwd_body<NKT, LK, PV16>(a, qt_per_block, L % nb, (L / nb) % a.H, L / (nb * a.H));
}

// Two independent attention problems in ONE launch (lmv_attn_fwd_pair): the image-token and the meta-token self-attention of an S
// block (models/lemevit.py:632,634).  The 16-token problem is B * h tiny workgroups -- ~12 us as a launch of its own, mostly ramp and
// tail -- and rides as extra workgroups behind the image-token ones here.
template <int NKT1, int LK1, int NKT2, int LK2>
__global__ __launch_bounds__(256) void mfma_fwd_pair_kernel(const AttnArgs a1, const AttnArgs a2, int per1, int per2, int nblk1) {
  // 1-D grid, all workgroups of problem 1 first: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so a 3-D grid
  // with the two problems interleaved along x would put every heavy workgroup on the even XCDs (measured: 52 vs 36 us)
  const int n1 = nblk1 * a1.H * a1.B;
  int id = (int)blockIdx.x;
  if (id < n1 && LMV_XCD_REMAP) id = xcd_contiguous(id, n1);
  if (id < n1) { const int bx = id % nblk1, t = id / nblk1; mfma_fwd_body<NKT1, LK1, true>(a1, per1, bx, t % a1.H, t / a1.H); }
  else { const int t = id - n1; mfma_fwd_body<NKT2, LK2, true>(a2, per2, 0, t % a2.H, t / a2.H); }
}

// =============================================================================================
// backward, dQ: one 16-query tile per wave iteration, loop over 32-key blocks
// =============================================================================================
template <int NKT>
__global__ __launch_bounds__(256) void mfma_bwd_dq_kernel(const AttnArgs a, float* __restrict__ delta, int qt_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[NKT * 16 * 64], sV[NKT * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D;
  const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + h * D;
  const bf16_t* ob = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_bs + h * D;
  bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + b * a.q_bs + h * D;
  stage_rows2<256>(sK, reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D, a.k_rs, sV, reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D, a.v_rs, 0, NKT * 16, a.Lk, tid);
  __syncthreads();
  const int nqt = (a.Lq + 15) >> 4;
  const int qt_end = min(nqt, (int)(blockIdx.x + 1) * qt_per_block);
  const int g = lane >> 4;
  const int64_t bh = ((int64_t)b * a.H + h) * a.Lq;
  for (int qt = blockIdx.x * qt_per_block + wave; qt < qt_end; qt += 4) {
    const int q = qt * 16 + (lane & 15);
    const bool vq = q < a.Lq;
    const bf16x8_t qf = load_frag_global(qb, a.q_rs, q, a.Lq, lane);
    const bf16x8_t gf = load_frag_global(gb, a.o_rs, q, a.Lq, lane);
    const bf16x8_t of = load_frag_global(ob, a.o_rs, q, a.Lq, lane);
    const float lse = vq ? a.lse[bh + q] : 0.f;
    // delta = rowsum(dO * O): the lane's 8 d-values, then the 4 lane groups of the query; kept for the dK / dV kernel
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)gf[j] * (float)of[j];
    dl = group_sum4(dl);
    if (vq && g == 0) delta[bh + q] = dl;
    f32x4_t dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb2 = 0; kb2 < NKT / 2; ++kb2) {
      f32x4_t ds[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kt = 2 * kb2 + t;
        const f32x4_t s = MFMA(frag_n(sK, kt * 16, lane), qf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
        const f32x4_t dp = MFMA(frag_n(sV, kt * 16, lane), gf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = vq && (kt * 16 + g * 4 + r < a.Lk);
          const float p = ok ? __expf(s[r] * a.scale - lse) : 0.f;
          ds[t][r] = p * (dp[r] - dl) * a.scale;
        }
      }
      const bf16x8_t dsf = pack8(ds[0], ds[1]);
      dq0 = MFMA(frag_t(sK, kb2 * 32, kb2 * 32 + 16, 0, lane), dsf, dq0);
      dq1 = MFMA(frag_t(sK, kb2 * 32, kb2 * 32 + 16, 16, lane), dsf, dq1);
    }
    if (vq) {
      store8(dqb + (int64_t)q * a.q_rs, g, dq0, dq1);
    }
  }
}

// =============================================================================================
// backward, dK / dV: one workgroup per (b, h, query range); wave w owns key tiles kt = w % KW, + KW, ... and the
// 32-query blocks qb = w / KW, + QW, ...  (KW * QW = 4).  Partial tiles of the QW query groups are summed in LDS.
// ATOMIC (name kept): several query ranges per (b, h) -> one fp32 partial slab per range, acc_k / acc_v [range][B][H][NKT*16][32],
// summed and scattered by scatter_sum_kernel (the first version used fp32 atomics: 5.5 M of them per launch at stage 1).
// =============================================================================================
constexpr int QR_MAX = 448;   // queries staged per workgroup (Q and dO images: 2 * 448 * 64 B = 56 KB)

// FUSEDQ (<= 32 keys, every wave holds both key tiles): the SAME kernel also produces dQ -- the D[key][q] orientation is
// recomputed from the fragments already in registers (two more S / dP MFMAs and exponentials per 16 queries, nothing against
// the memory traffic), delta = rowsum(dO * O) is formed while the query range is staged, and dQ rows leave complete (all keys
// of a (b, h) live in this workgroup).  Q, dO and O are then read ONCE by the backward pass instead of twice, and the
// separate dQ launch disappears (DCA x-direction: 3136 queries x 16 keys; the 16 x 16 meta-token self-attention).
// LK > 0: the key count is a compile-time constant (16 meta tokens: the x direction of Dual Cross-Attention): key tiles past the end are
// not computed at all -- with 16 keys in a 32-key block that is half of the MFMAs, exponentials and masks of the run-time version.
template <int NKT, int KW, bool ATOMIC, bool FUSEDQ = false, int LK = 0, int QRM = QR_MAX>
__device__ __forceinline__ void mfma_bwd_dkv_body(const AttnArgs& a, const float* __restrict__ delta, float* __restrict__ acc_k,
                                                  float* __restrict__ acc_v, int q_per_block, const int bx, const int h, const int b) {
  constexpr int QW = 4 / KW;
  constexpr int TPW = (NKT + KW - 1) / KW;     // key tiles per wave
  const int Lk = LK ? LK : a.Lk;
  static_assert(!FUSEDQ || (NKT == 2 && KW == 1), "the fused dQ path is written for two key tiles per wave");
  __shared__ __attribute__((aligned(16))) unsigned char sQG[2 * QRM * 64];     // Q image | dO image (reused by the final reduce)
  __shared__ __attribute__((aligned(16))) float sL[QRM], sDl[QRM];
  __shared__ __attribute__((aligned(16))) unsigned char sKV[FUSEDQ ? 2 * 32 * 64 : 16];                // FUSEDQ: K image | V image
  unsigned char* sQ = sQG;
  unsigned char* sG = sQG + QRM * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kw = wave % KW, qw = wave / KW, g = lane >> 4;
  const int q0 = bx * q_per_block, q1 = min(a.Lq, q0 + q_per_block);
  const int nrows = ((q1 - q0 + 31) >> 5) << 5;
  const int64_t bh = ((int64_t)b * a.H + h) * a.Lq;
  stage_rows2<256>(sQ, reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D, a.q_rs, sG, reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + h * D, a.o_rs, q0, nrows, q1, tid);
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  for (int i = tid; i < nrows; i += 256) {
    const bool ok = q0 + i < q1;
    sL[i] = ok ? a.lse[bh + q0 + i] : 1e30f;      // exp(s - 1e30) = 0 masks the padded queries
    if constexpr (FUSEDQ) {
      float dl = 0.f;                             // delta = rowsum(dO * O) of query q0 + i
      if (ok) {
        const bf16_t* gp = reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + (int64_t)(q0 + i) * a.o_rs + h * D;
        const bf16_t* op = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_bs + (int64_t)(q0 + i) * a.o_rs + h * D;
        uint4 rg[4], ro[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { rg[c] = *reinterpret_cast<const uint4*>(gp + c * 8); ro[c] = *reinterpret_cast<const uint4*>(op + c * 8); }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float gv[8], ov[8];
          chunk_to_f<bf16_t>(rg[c], gv); chunk_to_f<bf16_t>(ro[c], ov);
#pragma unroll
          for (int j = 0; j < 8; ++j) dl += gv[j] * ov[j];
        }
      }
      sDl[i] = dl;
    } else {
      sDl[i] = ok ? delta[bh + q0 + i] : 0.f;
    }
  }
  if constexpr (FUSEDQ) stage_rows2<256>(sKV, kb, a.k_rs, sKV + 32 * 64, vb, a.v_rs, 0, 32, Lk, tid);
  bf16x8_t kf[TPW], vf[TPW];
  f32x4_t dk[TPW][2], dv[TPW][2];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = kw + i * KW;
    kf[i] = load_frag_global(kb, a.k_rs, kt * 16 + (lane & 15), (kt < NKT) ? Lk : 0, lane);
    vf[i] = load_frag_global(vb, a.v_rs, kt * 16 + (lane & 15), (kt < NKT) ? Lk : 0, lane);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { dk[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  }
  __syncthreads();
  const bool vkey = true;
  for (int qb = qw; qb * 32 < nrows; qb += QW) {
    const int r0 = qb * 32;
    const bf16x8_t qn0 = frag_n(sQ, r0, lane), qn1 = frag_n(sQ, r0 + 16, lane);
    const bf16x8_t gn0 = frag_n(sG, r0, lane), gn1 = frag_n(sG, r0 + 16, lane);
    const bf16x8_t qt0 = frag_t(sQ, r0, r0 + 16, 0, lane), qt1 = frag_t(sQ, r0, r0 + 16, 16, lane);
    const bf16x8_t gt0 = frag_t(sG, r0, r0 + 16, 0, lane), gt1 = frag_t(sG, r0, r0 + 16, 16, lane);
    const float4 l0 = *reinterpret_cast<const float4*>(sL + r0 + g * 4), l1 = *reinterpret_cast<const float4*>(sL + r0 + 16 + g * 4);
    const float4 d0 = *reinterpret_cast<const float4*>(sDl + r0 + g * 4), d1 = *reinterpret_cast<const float4*>(sDl + r0 + 16 + g * 4);
    const float lse[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    const float dl[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    if constexpr (FUSEDQ) {
      // D[key][q]: lane holds keys kt * 16 + g * 4 + r of query (lane & 15) -- the orientation dQ = dS K needs
      bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + b * a.q_bs + h * D;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qi = r0 + t * 16 + (lane & 15);
        const float lq = sL[qi], dq_ = sDl[qi];
        const bf16x8_t qn = t ? qn1 : qn0, gn = t ? gn1 : gn0;
        f32x4_t dsq[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          if (LK && kt * 16 >= LK) { dsq[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; continue; }      // compile-time: padding tile
          const f32x4_t sv = MFMA(kf[kt], qn, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
          const f32x4_t dp = MFMA(vf[kt], gn, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool kin = (LK && kt * 16 + 16 <= LK) ? true : (kt * 16 + g * 4 + r < Lk);
            const float pq = kin ? __expf(sv[r] * a.scale - lq) : 0.f;
            dsq[kt][r] = pq * (dp[r] - dq_) * a.scale;
          }
        }
        const bf16x8_t dsf = pack8(dsq[0], dsq[1]);
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t dq0 = MFMA(frag_t(sKV, 0, 16, 0, lane), dsf, z), dq1 = MFMA(frag_t(sKV, 0, 16, 16, lane), dsf, z);
        if (q0 + qi < q1) {
          store8(dqb + (int64_t)(q0 + qi) * a.q_rs, g, dq0, dq1);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int kt = kw + i * KW;
      if (kt >= NKT) continue;
      if (LK && KW == 1 && kt * 16 >= LK) continue;                  // compile-time (KW == 1: kt = i): padding tile
      const bool kvalid = (LK && KW == 1 && kt * 16 + 16 <= LK) ? true : (vkey && (kt * 16 + (lane & 15) < Lk));
      // D[q][key]: lane holds queries r0 + t*16 + g*4 + r of key (lane & 15)
      const f32x4_t s0 = MFMA(qn0, kf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f})), s1 = MFMA(qn1, kf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      const f32x4_t p0 = MFMA(gn0, vf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f})), p1 = MFMA(gn1, vf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      f32x4_t pr[2], ds[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e0 = kvalid ? __expf(s0[r] * a.scale - lse[r]) : 0.f, e1 = kvalid ? __expf(s1[r] * a.scale - lse[4 + r]) : 0.f;
        pr[0][r] = e0; pr[1][r] = e1;
        ds[0][r] = e0 * (p0[r] - dl[r]) * a.scale; ds[1][r] = e1 * (p1[r] - dl[4 + r]) * a.scale;
      }
      const bf16x8_t pf = pack8(pr[0], pr[1]), dsf = pack8(ds[0], ds[1]);
      // D[d][key]: lane holds d = dt*16 + g*4 + r of key (lane & 15)
      dv[i][0] = MFMA(gt0, pf, dv[i][0]); dv[i][1] = MFMA(gt1, pf, dv[i][1]);
      dk[i][0] = MFMA(qt0, dsf, dk[i][0]); dk[i][1] = MFMA(qt1, dsf, dk[i][1]);
    }
  }
  // ---- combine the QW query groups through LDS (reuses the Q image), then write / accumulate ----------
  if (QW > 1) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(sQG);         // [wave][TPW][4 frags][64 lanes][4] <= 32 KB
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        *reinterpret_cast<f32x4_t*>(red + ((((qw * KW + kw) * TPW + i) * 4 + dt) * 64 + lane) * 4) = dk[i][dt];
        *reinterpret_cast<f32x4_t*>(red + ((((qw * KW + kw) * TPW + i) * 4 + 2 + dt) * 64 + lane) * 4) = dv[i][dt];
      }
    __syncthreads();
    if (qw != 0) return;
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int o = 1; o < QW; ++o) {
          dk[i][dt] += *reinterpret_cast<const f32x4_t*>(red + ((((o * KW + kw) * TPW + i) * 4 + dt) * 64 + lane) * 4);
          dv[i][dt] += *reinterpret_cast<const f32x4_t*>(red + ((((o * KW + kw) * TPW + i) * 4 + 2 + dt) * 64 + lane) * 4);
        }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = kw + i * KW, key = kt * 16 + (lane & 15);
    if (kt >= NKT || key >= Lk) continue;
    if (ATOMIC) {      // several query ranges per (b, h): this range's partial goes to ITS slab (plain stores; summed by scatter_sum_kernel)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int d = g * 8 + dt * 4;          // frag_t's interleaved column order
        const int64_t o = ((((int64_t)bx * a.B + b) * a.H + h) * (NKT * 16) + key) * D + d;
        *reinterpret_cast<f32x4_t*>(acc_k + o) = dk[i][dt];
        *reinterpret_cast<f32x4_t*>(acc_v + o) = dv[i][dt];
      }
    } else {
      store8(reinterpret_cast<bf16_t*>(a.dk) + b * a.k_bs + (int64_t)key * a.k_rs + h * D, g, dk[i][0], dk[i][1]);
      store8(reinterpret_cast<bf16_t*>(a.dv) + b * a.v_bs + (int64_t)key * a.v_rs + h * D, g, dv[i][0], dv[i][1]);
    }
  }
}

template <int NKT, int KW, bool ATOMIC, bool FUSEDQ = false, int LK = 0>
__global__ __launch_bounds__(256) void mfma_bwd_dkv_kernel(const AttnArgs a, const float* __restrict__ delta, float* __restrict__ acc_k,
                                                          float* __restrict__ acc_v, int q_per_block) {
  const int nb = gridDim.x, n = nb * a.H * a.B, id = (int)blockIdx.x + nb * ((int)blockIdx.y + a.H * (int)blockIdx.z);
  const int L = LMV_XCD_REMAP ? xcd_contiguous(id, n) : id;
  mfma_bwd_dkv_body<NKT, KW, ATOMIC, FUSEDQ, LK>(a, delta, acc_k, acc_v, q_per_block, L % nb, (L / nb) % a.H, L / (nb * a.H));
}

// =============================================================================================
// backward, FUSED (round 2): ONE workgroup per (b, h) produces dQ, dK and dV for Lq, Lk <= 224 (the stage-3 / stage-4
// self-attention).  Before, mfma_bwd_dq_long_kernel and mfma_bwd_dkv_kernel each staged their operands, recomputed S and dP and
// exponentiated every score -- and the kernels are VALU-issue bound, so the exponentials were paid twice.  Here:
//   * wave w owns the key tiles w, w + 4, ... (dK / dV accumulators in registers over ALL queries: no cross-wave reduce for them);
//   * per 32-query block and key tile, S and dP come out ONCE in the D[q][key] orientation; P and dS (bf16) feed the dV / dK MFMAs
//     directly as B operands, and dS is ALSO written to a wave-private 16 x 16 LDS tile [key][q] whose ds_read_b64_tr_b16 read is
//     exactly the B operand dQ^T = K^T dS^T needs (k-slots = keys, lane = query): a hardware transpose instead of a recompute;
//   * the four waves' dQ partials of a query block (each over its own key tiles) meet in LDS and leave in a fixed order.
// LK > 0: compile-time key count (196 / 49): padding tiles are skipped, only the boundary tile is masked.
// LDS: Q, dO, K images (<= 14 KB each) + 16 KB transposers + 16 KB dQ exchange = 77 KB: two workgroups per CU.
// =============================================================================================
template <int NKT, int LK = 0>
__device__ __forceinline__ void mfma_bwd_fused_body(const AttnArgs& a, const int h, const int b) {
  constexpr int TPW = (NKT + 3) / 4;          // key tiles per wave
  constexpr int NP = (TPW + 1) / 2;           // 32-key pairs per wave for the dQ contraction
  constexpr int NQ = NKT * 16;                // query rows staged (Lq <= NQ is guaranteed by the dispatcher)
  __shared__ __attribute__((aligned(16))) unsigned char sQ[NQ * 64], sG[NQ * 64], sK[NQ * 64];
  __shared__ __attribute__((aligned(16))) float sL[NQ], sDl[NQ];
  // [wave][q tile][key tile slot][16 keys x 16 queries bf16] transposers; the SAME 4 KB per wave then carry that wave's dQ partial to
  // the exchange ([q tile * 2 + d half][lane][4] fp32): a wave has read its tiles before it overwrites them, nobody else touches them
  constexpr int TSZ = (2 * 2 * NP * 512 > 4096) ? 2 * 2 * NP * 512 : 4096;
  __shared__ __attribute__((aligned(16))) unsigned char sTR[4][TSZ];
  auto sT = [&](int w, int t, int slot) -> unsigned char* { return &sTR[w][(t * 2 * NP + slot) * 512]; };
  auto sRed = [&](int w, int idx) -> float* { return reinterpret_cast<float*>(&sTR[w][idx * 1024]); };
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int Lk = LK ? LK : a.Lk, Lq = a.Lq;
  const int nrows = ((Lq + 31) >> 5) << 5;
  const int64_t bh = ((int64_t)b * a.H + h) * Lq;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  stage_rows2<256>(sQ, reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D, a.q_rs, sG, reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + h * D, a.o_rs, 0, nrows, Lq, tid);
  stage_rows2<256>(sK, kb, a.k_rs, sK, kb, a.k_rs, 0, NKT * 16, Lk, tid);         // K image for the K^T fragments of dQ (second copy is the same store)
  for (int i = tid; i < nrows; i += 256) {
    const bool ok = i < Lq;
    sL[i] = ok ? a.lse[bh + i] : 1e30f;           // exp(s - 1e30) = 0 masks the padded queries
    float dl = 0.f;                               // delta = rowsum(dO * O) of query i
    if (ok) {
      const bf16_t* gp = reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D;
      const bf16_t* op = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D;
      uint4 rg[4], ro[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { rg[c] = *reinterpret_cast<const uint4*>(gp + c * 8); ro[c] = *reinterpret_cast<const uint4*>(op + c * 8); }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float gv[8], ov[8];
        chunk_to_f<bf16_t>(rg[c], gv); chunk_to_f<bf16_t>(ro[c], ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += gv[j] * ov[j];
      }
    }
    sDl[i] = dl;
  }
  bf16x8_t kf[TPW], vf[TPW];
  f32x4_t dk[TPW][2], dv[TPW][2];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = wave + i * 4;
    kf[i] = load_frag_global(kb, a.k_rs, kt * 16 + (lane & 15), (kt < NKT) ? Lk : 0, lane);
    vf[i] = load_frag_global(vb, a.v_rs, kt * 16 + (lane & 15), (kt < NKT) ? Lk : 0, lane);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { dk[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  }
  __syncthreads();
  bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + b * a.q_bs + h * D;
  for (int r0 = 0; r0 < nrows; r0 += 32) {
    const bf16x8_t qn0 = frag_n(sQ, r0, lane), qn1 = frag_n(sQ, r0 + 16, lane);
    const bf16x8_t gn0 = frag_n(sG, r0, lane), gn1 = frag_n(sG, r0 + 16, lane);
    const bf16x8_t qt0 = frag_t(sQ, r0, r0 + 16, 0, lane), qt1 = frag_t(sQ, r0, r0 + 16, 16, lane);
    const bf16x8_t gt0 = frag_t(sG, r0, r0 + 16, 0, lane), gt1 = frag_t(sG, r0, r0 + 16, 16, lane);
    const float4 l0 = *reinterpret_cast<const float4*>(sL + r0 + g * 4), l1 = *reinterpret_cast<const float4*>(sL + r0 + 16 + g * 4);
    const float4 d0 = *reinterpret_cast<const float4*>(sDl + r0 + g * 4), d1 = *reinterpret_cast<const float4*>(sDl + r0 + 16 + g * 4);
    const float lse[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    const float dl[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int kt = wave + i * 4;
      if (kt >= NKT || kt * 16 >= Lk) continue;                      // wave-uniform: tile past the end
      const bool kvalid = kt * 16 + (lane & 15) < Lk;
      // D[q][key]: lane holds queries r0 + t * 16 + g * 4 + r of key (lane & 15)
      const f32x4_t s0 = MFMA(qn0, kf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f})), s1 = MFMA(qn1, kf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      const f32x4_t p0 = MFMA(gn0, vf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f})), p1 = MFMA(gn1, vf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      f32x4_t pr[2], ds[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e0 = kvalid ? __expf(s0[r] * a.scale - lse[r]) : 0.f, e1 = kvalid ? __expf(s1[r] * a.scale - lse[4 + r]) : 0.f;
        pr[0][r] = e0; pr[1][r] = e1;
        ds[0][r] = e0 * (p0[r] - dl[r]) * a.scale; ds[1][r] = e1 * (p1[r] - dl[4 + r]) * a.scale;
      }
      const bf16x8_t pf = pack8(pr[0], pr[1]), dsf = pack8(ds[0], ds[1]);
      dv[i][0] = MFMA(gt0, pf, dv[i][0]); dv[i][1] = MFMA(gt1, pf, dv[i][1]);
      dk[i][0] = MFMA(qt0, dsf, dk[i][0]); dk[i][1] = MFMA(qt1, dsf, dk[i][1]);
      // dS tiles -> the wave's transposers: row = key, 4 consecutive queries (8 bytes) at column g * 4
      const uint4 w = __builtin_bit_cast(uint4, dsf);
      *reinterpret_cast<uint2*>(sT(wave, 0, i) + (lane & 15) * 32 + g * 8) = make_uint2(w.x, w.y);
      *reinterpret_cast<uint2*>(sT(wave, 1, i) + (lane & 15) * 32 + g * 8) = make_uint2(w.z, w.w);
    }
    // ---- this wave's share of dQ^T[d][q] = sum over its keys of K^T[d][key] dS^T[key][q] ------------------------------------
    f32x4_t dq[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { dq[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dq[t][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    {
      const int rr = (lane & 15) >> 2, qq = lane & 3;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const int kta = wave + (2 * pp) * 4, ktb = wave + (2 * pp + 1) * 4;       // the two key tiles of this 32-key pair
        if (kta >= NKT || kta * 16 >= Lk) continue;
        const bool has_b = 2 * pp + 1 < TPW && ktb < NKT && ktb * 16 < Lk;          // wave-uniform: a missing second tile contributes zeros
        const int rowb = has_b ? ktb * 16 : kta * 16;
        const bf16x8_t ka = frag_t(sK, kta * 16, rowb, 0, lane), kb2 = frag_t(sK, kta * 16, rowb, 16, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sT(wave, t, 2 * pp) + (g * 4 + rr) * 32 + qq * 8));
          bf16x4_t hi = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
          if (has_b) hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sT(wave, t, 2 * pp + (2 * pp + 1 < TPW ? 1 : 0)) + (g * 4 + rr) * 32 + qq * 8));
          const bf16x8_t dst = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          dq[t][0] = MFMA(ka, dst, dq[t][0]); dq[t][1] = MFMA(kb2, dst, dq[t][1]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) *reinterpret_cast<f32x4_t*>(sRed(wave, t * 2 + hh) + lane * 4) = dq[t][hh];
    __syncthreads();
    if (wave < 2) {                               // wave t sums the four partials of query tile t in wave order and stores the rows
      f32x4_t x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        x0 += *reinterpret_cast<const f32x4_t*>(sRed(w, wave * 2) + lane * 4);
        x1 += *reinterpret_cast<const f32x4_t*>(sRed(w, wave * 2 + 1) + lane * 4);
      }
      const int q = r0 + wave * 16 + (lane & 15);
      if (q < Lq) store8(dqb + (int64_t)q * a.q_rs, g, x0, x1);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = wave + i * 4, key = kt * 16 + (lane & 15);
    if (kt >= NKT || key >= Lk) continue;
    store8(reinterpret_cast<bf16_t*>(a.dk) + b * a.k_bs + (int64_t)key * a.k_rs + h * D, g, dk[i][0], dk[i][1]);
    store8(reinterpret_cast<bf16_t*>(a.dv) + b * a.v_bs + (int64_t)key * a.v_rs + h * D, g, dv[i][0], dv[i][1]);
  }
}


// ---- round 5: the same fused backward with the dQ contraction split by QUERY TILE instead of by key tile ---------------------------------------------------
// In mfma_bwd_fused_body every wave forms a dQ partial over ITS key tiles for both query tiles of a 32-query block, the four partials meet in LDS (fp32) and two waves reduce and
// store them: two barriers per block, 16 KB of exchange, and the dQ work follows the 4 : 3 : 3 : 3 split of 13 key tiles.  Here, after ONE barrier (every wave's dS tiles of the
// block are in the transposers), wave w computes the COMPLETE dQ^T tile (query tile w >> 1, d half w & 1) over all key tiles -- 7 MFMAs on 7 K^T fragments it loaded once and keeps
// in registers (the K image is then dead: its LDS is the second transposer buffer) -- and stores its 8 bytes per lane itself: no partials, no reduce, balanced, and with the
// transposers double-buffered by block parity the second barrier goes too (a wave can only reach block j + 2's writes of a buffer behind barrier j + 1, which every wave passes
// after its reads of block j).  train step: main-stream time is worth ~1.2 ms per ms here (LMV_DBG_SKIP_ATTN_BWD: -3.3 ms for 2.75 ms of launches).
template <int NKT, int LK = 0>
__device__ __forceinline__ void mfma_bwd_fused_body2(const AttnArgs& a, const int h, const int b) {
  constexpr int TPW = (NKT + 3) / 4;          // key tiles per wave (scores / dK / dV: by key tile, as before)
  constexpr int NPAIR = (NKT + 1) / 2;        // 32-key steps of the dQ contraction
  constexpr int NQ = NKT * 16;
  constexpr int TBUF = 4 * 2 * TPW * 512;     // one transposer buffer: [wave][q tile][local key tile][16 keys x 16 queries bf16]
  __shared__ __attribute__((aligned(16))) unsigned char sQ[NQ * 64], sG[NQ * 64];
  __shared__ __attribute__((aligned(16))) unsigned char sKT[(2 * TBUF > NQ * 64) ? 2 * TBUF : NQ * 64];      // K image during the prologue, then both transposer buffers
  __shared__ __attribute__((aligned(16))) float sL[NQ], sDl[NQ];
  auto sT = [&](int buf, int w, int t, int slot) -> unsigned char* { return &sKT[buf * TBUF + ((w * 2 + t) * TPW + slot) * 512]; };
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int Lk = LK ? LK : a.Lk, Lq = a.Lq;
  const int nrows = ((Lq + 31) >> 5) << 5;
  const int64_t bh = ((int64_t)b * a.H + h) * Lq;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  stage_rows2<256>(sQ, reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D, a.q_rs, sG, reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + h * D, a.o_rs, 0, nrows, Lq, tid);
  stage_rows2<256>(sKT, kb, a.k_rs, sKT, kb, a.k_rs, 0, NKT * 16, Lk, tid);
  for (int i = tid; i < nrows; i += 256) {
    const bool ok = i < Lq;
    sL[i] = ok ? a.lse[bh + i] : 1e30f;
    float dl = 0.f;
    if (ok) {
      const bf16_t* gp = reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D;
      const bf16_t* op = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_bs + (int64_t)i * a.o_rs + h * D;
      uint4 rg[4], ro[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { rg[c] = *reinterpret_cast<const uint4*>(gp + c * 8); ro[c] = *reinterpret_cast<const uint4*>(op + c * 8); }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float gv[8], ov[8];
        chunk_to_f<bf16_t>(rg[c], gv); chunk_to_f<bf16_t>(ro[c], ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += gv[j] * ov[j];
      }
    }
    sDl[i] = dl;
  }
  bf16x8_t kf[TPW], vf[TPW];
  f32x4_t dk[TPW][2], dv[TPW][2];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = wave + i * 4;
    kf[i] = load_frag_global(kb, a.k_rs, kt * 16 + (lane & 15), (kt < NKT) ? Lk : 0, lane);
    vf[i] = load_frag_global(vb, a.v_rs, kt * 16 + (lane & 15), (kt < NKT) ? Lk : 0, lane);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { dk[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  }
  __syncthreads();
  // this wave's K^T fragments of the dQ contraction: d half (wave & 1), key tiles (2 pp, 2 pp + 1); rows past Lk are zero in the image
  const int qt_w = wave >> 1, dh = wave & 1;
  bf16x8_t ktf[NPAIR];
#pragma unroll
  for (int pp = 0; pp < NPAIR; ++pp) {
    const int rowb = (2 * pp + 1 < NKT) ? (2 * pp + 1) * 16 : 2 * pp * 16;
    ktf[pp] = frag_t(sKT, 2 * pp * 16, rowb, dh * 16, lane);
  }
  __syncthreads();          // the K image is dead from here: its bytes are the transposer buffers
  bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + b * a.q_bs + h * D;
  int buf = 0;
  for (int r0 = 0; r0 < nrows; r0 += 32, buf ^= 1) {
    const bf16x8_t qn0 = frag_n(sQ, r0, lane), qn1 = frag_n(sQ, r0 + 16, lane);
    const bf16x8_t gn0 = frag_n(sG, r0, lane), gn1 = frag_n(sG, r0 + 16, lane);
    const bf16x8_t qt0 = frag_t(sQ, r0, r0 + 16, 0, lane), qt1 = frag_t(sQ, r0, r0 + 16, 16, lane);
    const bf16x8_t gt0 = frag_t(sG, r0, r0 + 16, 0, lane), gt1 = frag_t(sG, r0, r0 + 16, 16, lane);
    const float4 l0 = *reinterpret_cast<const float4*>(sL + r0 + g * 4), l1 = *reinterpret_cast<const float4*>(sL + r0 + 16 + g * 4);
    const float4 d0 = *reinterpret_cast<const float4*>(sDl + r0 + g * 4), d1 = *reinterpret_cast<const float4*>(sDl + r0 + 16 + g * 4);
    const float lse[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    const float dl[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int kt = wave + i * 4;
      if (kt >= NKT || kt * 16 >= Lk) continue;                      // wave-uniform: tile past the end
      const bool kvalid = kt * 16 + (lane & 15) < Lk;
      const f32x4_t s0 = MFMA(qn0, kf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f})), s1 = MFMA(qn1, kf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      const f32x4_t p0 = MFMA(gn0, vf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f})), p1 = MFMA(gn1, vf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      f32x4_t pr[2], ds[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e0 = kvalid ? __expf(s0[r] * a.scale - lse[r]) : 0.f, e1 = kvalid ? __expf(s1[r] * a.scale - lse[4 + r]) : 0.f;
        pr[0][r] = e0; pr[1][r] = e1;
        ds[0][r] = e0 * (p0[r] - dl[r]) * a.scale; ds[1][r] = e1 * (p1[r] - dl[4 + r]) * a.scale;
      }
      const bf16x8_t pf = pack8(pr[0], pr[1]), dsf = pack8(ds[0], ds[1]);
      dv[i][0] = MFMA(gt0, pf, dv[i][0]); dv[i][1] = MFMA(gt1, pf, dv[i][1]);
      dk[i][0] = MFMA(qt0, dsf, dk[i][0]); dk[i][1] = MFMA(qt1, dsf, dk[i][1]);
      const uint4 w = __builtin_bit_cast(uint4, dsf);
      // (8-byte piece g of key row (lane & 15) sits at position g ^ (row >> 2): the 16 lanes of a ds_write_b64 lane group share g, and with the pieces in place their rows hit 4
      //  banks -- 16 LDS cycles per store instead of 4, 60 % of this loop's LDS time (tools/lds_conflicts.py); the transposing read below undoes it in its address)
      const int wsl = (g ^ ((lane >> 2) & 3)) * 8;
      *reinterpret_cast<uint2*>(sT(buf, wave, 0, i) + (lane & 15) * 32 + wsl) = make_uint2(w.x, w.y);
      *reinterpret_cast<uint2*>(sT(buf, wave, 1, i) + (lane & 15) * 32 + wsl) = make_uint2(w.z, w.w);
    }
    __syncthreads();          // every key tile's dS of this block is in buffer `buf`
    {
      const int rr = (lane & 15) >> 2, qq = lane & 3;
      f32x4_t dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pp = 0; pp < NPAIR; ++pp) {
        const int kta = 2 * pp, ktb = 2 * pp + 1;
        if (kta * 16 >= Lk) continue;
        const bool has_b = ktb < NKT && ktb * 16 < Lk;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sT(buf, kta & 3, qt_w, kta >> 2) + (g * 4 + rr) * 32 + (qq ^ g) * 8));          // (row >> 2 = g)
        bf16x4_t hi = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        if (has_b) hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(sT(buf, ktb & 3, qt_w, ktb >> 2) + (g * 4 + rr) * 32 + (qq ^ g) * 8));
        const bf16x8_t dst = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        dq = MFMA(ktf[pp], dst, dq);
      }
      const int q = r0 + qt_w * 16 + (lane & 15);
      if (q < Lq) store4(dqb + (int64_t)q * a.q_rs + g * 8 + dh * 4, dq);          // (frag_t's d order: half dh of the lane's 8 consecutive d)
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = wave + i * 4, key = kt * 16 + (lane & 15);
    if (kt >= NKT || key >= Lk) continue;
    store8(reinterpret_cast<bf16_t*>(a.dk) + b * a.k_bs + (int64_t)key * a.k_rs + h * D, g, dk[i][0], dk[i][1]);
    store8(reinterpret_cast<bf16_t*>(a.dv) + b * a.v_bs + (int64_t)key * a.v_rs + h * D, g, dv[i][0], dv[i][1]);
  }
}

template <int NKT, int LK = 0, int VER = 1>
__global__ __launch_bounds__(256, 2) void mfma_bwd_fused_kernel(const AttnArgs a) {
  const int n = a.H * a.B, id = (int)blockIdx.y + a.H * (int)blockIdx.z;
  const int L = LMV_XCD_REMAP ? xcd_contiguous(id, n) : id;
  if constexpr (VER == 2) mfma_bwd_fused_body2<NKT, LK>(a, L % a.H, L / a.H);
  else mfma_bwd_fused_body<NKT, LK>(a, L % a.H, L / a.H);
}

// backward of the pair launch: workgroup (0, h, b) differentiates the image-token problem (fused kernel), workgroup (1, h, b) the
// 16-token problem (the <= 32-key kernel that also produces dQ, with 32-row staging images instead of 448-row ones)
template <int NKT1, int LK1, int VER = 1>
__global__ __launch_bounds__(256, 2) void mfma_bwd_pair_kernel(const AttnArgs a1, const AttnArgs a2) {
  const int n1 = a1.H * a1.B;                                 // 1-D grid, problem 1 first (see mfma_fwd_pair_kernel)
  int id = (int)blockIdx.x;
  if (id < n1 && LMV_XCD_REMAP) id = xcd_contiguous(id, n1);
  if (id < n1) { if constexpr (VER == 2) mfma_bwd_fused_body2<NKT1, LK1>(a1, id % a1.H, id / a1.H); else mfma_bwd_fused_body<NKT1, LK1>(a1, id % a1.H, id / a1.H); }
  else mfma_bwd_dkv_body<2, 1, false, true, 16, 32>(a2, nullptr, nullptr, nullptr, 32, 0, (id - n1) % a2.H, (id - n1) / a2.H);
}

// fp32 [B][H][LP][32] accumulator -> strided bf16 rows (first L rows)
__global__ __launch_bounds__(256) void scatter_bf16_kernel(const float* __restrict__ acc, bf16_t* __restrict__ dst, int64_t bs, int64_t rs, int B, int H,
                                                          int L, int LP) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x, total = (unsigned)B * H * L * D;
  if (idx >= total) return;
  const unsigned d = idx % D, l = (idx / D) % L, h = (idx / (D * L)) % H, b = idx / (D * L * H);
  dst[b * bs + (int64_t)l * rs + h * D + d] = f2bf(acc[((((int64_t)b * H + h) * LP) + l) * D + d]);
}

// dst[b][l][h][:] = sum over `nslab` slabs of acc[slab][b][h][l][:]  (two tensors per launch: dK and dV)
__global__ __launch_bounds__(256) void scatter_sum_kernel(const float* __restrict__ acc0, const float* __restrict__ acc1, bf16_t* __restrict__ dst0,
                                                         bf16_t* __restrict__ dst1, int64_t bs0, int64_t rs0, int64_t bs1, int64_t rs1, int B, int H, int L,
                                                         int LP, int nslab) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x, per = (unsigned)B * H * L * (D / 4);
  if (idx >= 2 * per) return;
  const bool second = idx >= per;
  const unsigned i = second ? idx - per : idx;
  const unsigned d4 = i % (D / 4), l = (i / (D / 4)) % L, h = (i / ((D / 4) * L)) % H, b = i / ((D / 4) * L * H);
  const float* src = (second ? acc1 : acc0) + ((((int64_t)b * H + h) * LP) + l) * D + d4 * 4;
  const int64_t slab = (int64_t)B * H * LP * D;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < nslab; ++k) s += *reinterpret_cast<const f32x4_t*>(src + k * slab);
  bf16_t* dst = second ? dst1 + b * bs1 + (int64_t)l * rs1 + h * D + d4 * 4 : dst0 + b * bs0 + (int64_t)l * rs0 + h * D + d4 * 4;
  store4(dst, s);
}

// =============================================================================================
// <= 16 queries over MANY keys (meta tokens attend to N image tokens: DCA c-direction, stage-0 CA).
// B * h alone under-fills 256 CUs, so the KEYS are split: one wavefront per range of RK = 128 keys (4 ranges per
// workgroup); forward writes per-range (max, sum, unnormalised O) partials that fwd_fewq_combine_kernel merges;
// backward is ONE kernel: every wave produces dK / dV of its own keys (D[q][key] orientation) and its share of dQ
// (D[key][q] orientation), the four waves' dQ are summed in LDS and added to an fp32 accumulator.
// =============================================================================================
constexpr int RKT = 8, RK = RKT * 16;      // forward: key tiles / keys per wave
constexpr int RKT_B = 4, RKB = RKT_B * 16; // backward (more LDS per key: Q / dO images, reduction scratch)

__global__ __launch_bounds__(256) void mfma_fwd_split_kernel(const AttnArgs a, float* __restrict__ part, int nsplit) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[4 * RK * 64], sV[4 * RK * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int nbx = gridDim.x, Lid = LMV_XCD_REMAP ? xcd_contiguous((int)blockIdx.x + nbx * ((int)blockIdx.y + a.H * (int)blockIdx.z), nbx * a.H * a.B)
                                                 : (int)blockIdx.x + nbx * ((int)blockIdx.y + a.H * (int)blockIdx.z);
  const int bx = Lid % nbx, h = (Lid / nbx) % a.H, b = Lid / (nbx * a.H);
  const int split = bx * 4 + wave, k0 = split * RK;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  unsigned char* wK = sK + wave * RK * 64;
  unsigned char* wV = sV + wave * RK * 64;
  stage_rows2<64>(wK, kb, a.k_rs, wV, vb, a.v_rs, k0, RK, a.Lk, lane);          // per-wave images: only wave-level ordering needed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (split >= nsplit) return;
  const int q = lane & 15;
  float* pp = part + ((((int64_t)b * a.H + h) * nsplit + split) * 16 + q) * (D + 2);
  if (k0 >= a.Lk) {                                              // empty range: neutral partial
    if (g == 0) { pp[0] = -1e30f; pp[1] = 0.f; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { pp[2 + g * 4 + r] = 0.f; pp[18 + g * 4 + r] = 0.f; }
    return;
  }
  const bf16x8_t qf = load_frag_global(reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D, a.q_rs, q, a.Lq, lane);
  f32x4_t s[RKT];
  float m = -1e30f;
#pragma unroll
  for (int kt = 0; kt < RKT; ++kt) {
    s[kt] = MFMA(frag_n(wK, kt * 16, lane), qf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[kt][r] = (k0 + kt * 16 + g * 4 + r < a.Lk) ? s[kt][r] * a.scale : -1e30f;
      m = fmaxf(m, s[kt][r]);
    }
  }
  m = group_max4(m);
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < RKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float p = __expf(s[kt][r] - m); s[kt][r] = p; l += p; }
  l = group_sum4(l);
  f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb2 = 0; kb2 < RKT / 2; ++kb2) {
    const bf16x8_t ph = pack8(s[2 * kb2], s[2 * kb2 + 1]);
    f32x4_t r0, r1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { r0[r] = s[2 * kb2][r] - (float)ph[r]; r1[r] = s[2 * kb2 + 1][r] - (float)ph[4 + r]; }
    const bf16x8_t pl = pack8(r0, r1);
    const bf16x8_t vt0 = frag_t(wV, kb2 * 32, kb2 * 32 + 16, 0, lane), vt1 = frag_t(wV, kb2 * 32, kb2 * 32 + 16, 16, lane);
    o0 = MFMA(vt0, ph, o0); o0 = MFMA(vt0, pl, o0);
    o1 = MFMA(vt1, ph, o1); o1 = MFMA(vt1, pl, o1);
  }
  if (g == 0) { pp[0] = m; pp[1] = l; }
#pragma unroll
  for (int r = 0; r < 4; ++r) { pp[2 + g * 8 + r] = o0[r]; pp[2 + g * 8 + 4 + r] = o1[r]; }      // frag_t's interleaved column order
}

// backward of the same shape, RKB = 64 keys per wave: grid (ceil(ceil(Lk / 64) / 4), H, B)
__global__ __launch_bounds__(256) void mfma_bwd_fewq_kernel(const AttnArgs a, float* __restrict__ acc_q) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[4 * RKB * 64], sV[4 * RKB * 64];
  __shared__ __attribute__((aligned(16))) unsigned char sQ[32 * 64], sG[32 * 64];
  __shared__ __attribute__((aligned(16))) float sL[32], sDl[32], sRed[4][2][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int nbx = gridDim.x, Lid = LMV_XCD_REMAP ? xcd_contiguous((int)blockIdx.x + nbx * ((int)blockIdx.y + a.H * (int)blockIdx.z), nbx * a.H * a.B)
                                                 : (int)blockIdx.x + nbx * ((int)blockIdx.y + a.H * (int)blockIdx.z);
  const int bx = Lid % nbx, h = (Lid / nbx) % a.H, b = Lid / (nbx * a.H);
  const int k0 = (bx * 4 + wave) * RKB;
  const int64_t bh = ((int64_t)b * a.H + h) * a.Lq;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  unsigned char* wK = sK + wave * RKB * 64;
  unsigned char* wV = sV + wave * RKB * 64;
  stage_rows2<64>(wK, kb, a.k_rs, wV, vb, a.v_rs, k0, RKB, a.Lk, lane);
  stage_rows2<256>(sQ, reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D, a.q_rs, sG, reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + h * D, a.o_rs, 0, 32, a.Lq, tid);
  if (tid < 32) {
    const bool ok = tid < a.Lq;
    sL[tid] = ok ? a.lse[bh + tid] : 1e30f;     // exp(s - 1e30) = 0 masks the padded queries
    float dl = 0.f;                             // delta = rowsum(dO * O) of query `tid`
    if (ok) {
      const bf16_t* gp = reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + (int64_t)tid * a.o_rs + h * D;
      const bf16_t* op = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_bs + (int64_t)tid * a.o_rs + h * D;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float gv[8], ov[8];
        chunk_to_f<bf16_t>(*reinterpret_cast<const uint4*>(gp + c * 8), gv);
        chunk_to_f<bf16_t>(*reinterpret_cast<const uint4*>(op + c * 8), ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += gv[j] * ov[j];
      }
    }
    sDl[tid] = dl;
  }
  __syncthreads();
  // ---- orientation D[q][key]: dK, dV of this wave's keys (query rows 0..31, rows >= Lq are zero / masked) --------
  const bf16x8_t qn0 = frag_n(sQ, 0, lane), qn1 = frag_n(sQ, 16, lane), gn0 = frag_n(sG, 0, lane), gn1 = frag_n(sG, 16, lane);
  const bf16x8_t qt0 = frag_t(sQ, 0, 16, 0, lane), qt1 = frag_t(sQ, 0, 16, 16, lane);
  const bf16x8_t gt0 = frag_t(sG, 0, 16, 0, lane), gt1 = frag_t(sG, 0, 16, 16, lane);
  const float4 l0 = *reinterpret_cast<const float4*>(sL + g * 4), l1 = *reinterpret_cast<const float4*>(sL + 16 + g * 4);
  const float4 d0 = *reinterpret_cast<const float4*>(sDl + g * 4), d1 = *reinterpret_cast<const float4*>(sDl + 16 + g * 4);
  const float lse[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
  const float dl[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
  for (int kt = 0; kt < RKT_B; ++kt) {
    const int key = k0 + kt * 16 + (lane & 15);
    if (k0 + kt * 16 >= a.Lk) break;                              // wave-uniform
    const bool kvalid = key < a.Lk;
    const bf16x8_t kf = frag_n(wK, kt * 16, lane), vf = frag_n(wV, kt * 16, lane);
    const f32x4_t s0 = MFMA(qn0, kf, (f32x4_t{0.f, 0.f, 0.f, 0.f})), s1 = MFMA(qn1, kf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
    const f32x4_t p0 = MFMA(gn0, vf, (f32x4_t{0.f, 0.f, 0.f, 0.f})), p1 = MFMA(gn1, vf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
    f32x4_t pr[2], ds[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e0 = kvalid ? __expf(s0[r] * a.scale - lse[r]) : 0.f, e1 = kvalid ? __expf(s1[r] * a.scale - lse[4 + r]) : 0.f;
      pr[0][r] = e0; pr[1][r] = e1;
      ds[0][r] = e0 * (p0[r] - dl[r]) * a.scale; ds[1][r] = e1 * (p1[r] - dl[4 + r]) * a.scale;
    }
    const bf16x8_t pf = pack8(pr[0], pr[1]), dsf = pack8(ds[0], ds[1]);
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    const f32x4_t dv0 = MFMA(gt0, pf, z), dv1 = MFMA(gt1, pf, z), dk0 = MFMA(qt0, dsf, z), dk1 = MFMA(qt1, dsf, z);
    if (kvalid) {
      store8(reinterpret_cast<bf16_t*>(a.dk) + b * a.k_bs + (int64_t)key * a.k_rs + h * D, g, dk0, dk1);
      store8(reinterpret_cast<bf16_t*>(a.dv) + b * a.v_bs + (int64_t)key * a.v_rs + h * D, g, dv0, dv1);
    }
  }
  // ---- orientation D[key][q]: this wave's share of dQ (query tile 0: rows 0..15) --------------------------------
  const int q = lane & 15;
  const bool vq = q < a.Lq;
  const float lq = sL[q], dq_ = sDl[q];
  f32x4_t dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb2 = 0; kb2 < RKT_B / 2; ++kb2) {
    if (k0 + kb2 * 32 >= a.Lk) break;
    f32x4_t ds[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kt = 2 * kb2 + t;
      const f32x4_t s = MFMA(frag_n(wK, kt * 16, lane), qn0, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      const f32x4_t dp = MFMA(frag_n(wV, kt * 16, lane), gn0, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = vq && (k0 + kt * 16 + g * 4 + r < a.Lk);
        const float p = ok ? __expf(s[r] * a.scale - lq) : 0.f;
        ds[t][r] = p * (dp[r] - dq_) * a.scale;
      }
    }
    const bf16x8_t dsf = pack8(ds[0], ds[1]);
    dq0 = MFMA(frag_t(wK, kb2 * 32, kb2 * 32 + 16, 0, lane), dsf, dq0);
    dq1 = MFMA(frag_t(wK, kb2 * 32, kb2 * 32 + 16, 16, lane), dsf, dq1);
  }
  *reinterpret_cast<f32x4_t*>(&sRed[wave][0][lane][0]) = dq0;
  *reinterpret_cast<f32x4_t*>(&sRed[wave][1][lane][0]) = dq1;
  __syncthreads();
  if (wave == 0 && vq) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      dq0 += *reinterpret_cast<const f32x4_t*>(&sRed[w][0][lane][0]);
      dq1 += *reinterpret_cast<const f32x4_t*>(&sRed[w][1][lane][0]);
    }
    // this workgroup's share of dQ goes to ITS slab (plain stores, no atomics: summed in a fixed order by scatter_sum1_kernel)
    float* dst = acc_q + ((((int64_t)bx * a.B + b) * a.H + h) * 16 + q) * D + g * 8;
    *reinterpret_cast<f32x4_t*>(dst) = dq0;
    *reinterpret_cast<f32x4_t*>(dst + 4) = dq1;
  }
}

// dst[b][l][h][:] = sum over `nslab` slabs of acc[slab][b][h][l][:] for l < L (dQ of the few-query backward)
__global__ __launch_bounds__(256) void scatter_sum1_kernel(const float* __restrict__ acc, bf16_t* __restrict__ dst, int64_t bs, int64_t rs, int B, int H, int L,
                                                          int LP, int nslab) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x, total = (unsigned)B * H * L * (D / 8);
  if (idx >= total) return;
  const unsigned d8 = idx % (D / 8), l = (idx / (D / 8)) % L, h = (idx / ((D / 8) * L)) % H, b = idx / ((D / 8) * L * H);
  const float* src = acc + ((((int64_t)b * H + h) * LP) + l) * D + d8 * 8;
  const int64_t slab = (int64_t)B * H * LP * D;
  f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < nslab; ++k) { s0 += *reinterpret_cast<const f32x4_t*>(src + k * slab); s1 += *reinterpret_cast<const f32x4_t*>(src + k * slab + 4); }
  store8(dst + b * bs + (int64_t)l * rs + h * D, (int)d8, s0, s1);
}

// =============================================================================================
// Many queries x 225..640 keys (self-attention at 384^2: 576 + 16 tokens).  K and V of one (b, h) still fit in LDS
// (<= 80 KB), so these are the whole-row kernels with run-time tile counts and dynamic LDS:
//   forward: TWO passes over the key tiles (row maximum first, then exp / sum / PV with S recomputed) -- the score row
//            no longer fits in registers, and a second S MFMA per tile is cheaper than an online-softmax rescale that
//            needs two cross-lane-group shuffles per block;
//   dQ:      the block loop of mfma_bwd_dq_kernel with a run-time bound;
//   dK / dV: one workgroup per range of 14 key tiles, looping over query chunks staged in LDS (no atomics).
// =============================================================================================
constexpr int LONG_MAX_NKT = 40;

__global__ __launch_bounds__(256) void mfma_fwd_long_kernel(const AttnArgs a, int qt_per_block, int nkt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_long[];
  unsigned char* sK = smem_long;
  unsigned char* sV = smem_long + nkt * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D;
  bf16_t* ob = reinterpret_cast<bf16_t*>(a.o) + b * a.o_bs + h * D;
  stage_rows2<256>(sK, reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D, a.k_rs, sV, reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D, a.v_rs, 0, nkt * 16, a.Lk, tid);
  __syncthreads();
  const int nqt = (a.Lq + 15) >> 4;
  const int qt_end = min(nqt, (int)(blockIdx.x + 1) * qt_per_block);
  const int g = lane >> 4;
  for (int qt = blockIdx.x * qt_per_block + wave; qt < qt_end; qt += 4) {
    const int q = qt * 16 + (lane & 15);
    const bf16x8_t qf = load_frag_global(qb, a.q_rs, q, a.Lq, lane);
    float m = -3e38f;                                      // maximum of the RAW scores (scale > 0), see mfma_fwd_kernel
    for (int kt = 0; kt < nkt; ++kt) {
      const f32x4_t s = MFMA(frag_n(sK, kt * 16, lane), qf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
      if (kt * 16 + 16 > a.Lk) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 16 + g * 4 + r < a.Lk) m = fmaxf(m, s[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, s[r]);
      }
    }
    m = group_max4(m);
    const float sc2 = a.scale * 1.4426950408889634f, mc = m * sc2;
    float l = 0.f;
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    for (int kb2 = 0; kb2 < nkt / 2; ++kb2) {
      f32x4_t p[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kt = 2 * kb2 + t;
        const f32x4_t s = MFMA(frag_n(sK, kt * 16, lane), qf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[t][r] = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -mc));
          if (kt * 16 + 16 > a.Lk && kt * 16 + g * 4 + r >= a.Lk) p[t][r] = 0.f;
          l += p[t][r];
        }
      }
      const bf16x8_t ph = pack8(p[0], p[1]);                      // hi + lo parts of P: see mfma_fwd_kernel
      f32x4_t r0, r1;
#pragma unroll
      for (int r = 0; r < 4; ++r) { r0[r] = p[0][r] - (float)ph[r]; r1[r] = p[1][r] - (float)ph[4 + r]; }
      const bf16x8_t pl = pack8(r0, r1);
      const bf16x8_t vt0 = frag_t(sV, kb2 * 32, kb2 * 32 + 16, 0, lane), vt1 = frag_t(sV, kb2 * 32, kb2 * 32 + 16, 16, lane);
      o0 = MFMA(vt0, ph, o0); o0 = MFMA(vt0, pl, o0);
      o1 = MFMA(vt1, ph, o1); o1 = MFMA(vt1, pl, o1);
    }
    l = group_sum4(l);
    if (q < a.Lq) {
      const float inv = 1.f / l;
      o0 *= inv; o1 *= inv;
      store8(ob + (int64_t)q * a.o_rs, g, o0, o1);
      if (a.lse && g == 0) a.lse[((int64_t)b * a.H + h) * a.Lq + q] = m * a.scale + __logf(l);
    }
  }
}

__global__ __launch_bounds__(256) void mfma_bwd_dq_long_kernel(const AttnArgs a, float* __restrict__ delta, int qt_per_block, int nkt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_long[];
  unsigned char* sK = smem_long;
  unsigned char* sV = smem_long + nkt * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D;
  const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + h * D;
  const bf16_t* ob = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_bs + h * D;
  bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + b * a.q_bs + h * D;
  stage_rows2<256>(sK, reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D, a.k_rs, sV, reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D, a.v_rs, 0, nkt * 16, a.Lk, tid);
  __syncthreads();
  const int nqt = (a.Lq + 15) >> 4;
  const int qt_end = min(nqt, (int)(blockIdx.x + 1) * qt_per_block);
  const int g = lane >> 4;
  const int64_t bh = ((int64_t)b * a.H + h) * a.Lq;
  // operands of the NEXT query tile are fetched while this one computes (see mfma_fwd_kernel)
  const int qfirst = (blockIdx.x * qt_per_block + wave) * 16 + (lane & 15);
  bf16x8_t qf_n = load_frag_global(qb, a.q_rs, qfirst, a.Lq, lane), gf_n = load_frag_global(gb, a.o_rs, qfirst, a.Lq, lane);
  bf16x8_t of_n = load_frag_global(ob, a.o_rs, qfirst, a.Lq, lane);
  float lse_n = qfirst < a.Lq ? a.lse[bh + qfirst] : 0.f;
  for (int qt = blockIdx.x * qt_per_block + wave; qt < qt_end; qt += 4) {
    const int q = qt * 16 + (lane & 15);
    const bool vq = q < a.Lq;
    const bf16x8_t qf = qf_n, gf = gf_n, of = of_n;
    const float lse = lse_n;
    if (qt + 4 < qt_end) {
      qf_n = load_frag_global(qb, a.q_rs, q + 64, a.Lq, lane); gf_n = load_frag_global(gb, a.o_rs, q + 64, a.Lq, lane);
      of_n = load_frag_global(ob, a.o_rs, q + 64, a.Lq, lane);
      lse_n = q + 64 < a.Lq ? a.lse[bh + q + 64] : 0.f;
    }
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)gf[j] * (float)of[j];
    dl = group_sum4(dl);
    if (vq && g == 0) delta[bh + q] = dl;
    f32x4_t dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
    for (int kb2 = 0; kb2 < nkt / 2; ++kb2) {
      f32x4_t ds[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kt = 2 * kb2 + t;
        const f32x4_t s = MFMA(frag_n(sK, kt * 16, lane), qf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
        const f32x4_t dp = MFMA(frag_n(sV, kt * 16, lane), gf, (f32x4_t{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = vq && (kt * 16 + g * 4 + r < a.Lk);
          const float p = ok ? __expf(s[r] * a.scale - lse) : 0.f;
          ds[t][r] = p * (dp[r] - dl) * a.scale;
        }
      }
      const bf16x8_t dsf = pack8(ds[0], ds[1]);
      dq0 = MFMA(frag_t(sK, kb2 * 32, kb2 * 32 + 16, 0, lane), dsf, dq0);
      dq1 = MFMA(frag_t(sK, kb2 * 32, kb2 * 32 + 16, 16, lane), dsf, dq1);
    }
    if (vq) {
      store8(dqb + (int64_t)q * a.q_rs, g, dq0, dq1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// More than LONG_MAX_NKT * 16 keys (the dense-prediction backbones: stage-3 self-attention over 4096 image tokens at 1024^2;
// `vit_tiny` stage 0): K / V no longer fit in LDS, so they are STREAMED through it in chunks of STREAM_NKT key tiles (32 KB)
// while each wave keeps the state of up to STREAM_MAXQ query tiles in registers.
//   forward: online softmax across chunks -- per chunk the two passes of mfma_fwd_long_kernel (maximum, then P and P V) with
//            the running maximum, O and the row sum rescaled by 2^((m_old - m_new) c) when the maximum moves;
//   dQ:      dQ accumulates over the chunks (LSE and delta are chunk-independent);
//   dK / dV: mfma_bwd_dkv_long_kernel as it is (one workgroup per 14 key tiles, queries streamed).
// ---------------------------------------------------------------------------------------------
constexpr int STREAM_NKT = 16, STREAM_MAXQ = 4;

__global__ __launch_bounds__(256) void mfma_fwd_stream_kernel(const AttnArgs a, int qt_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[STREAM_NKT * 1024], sV[STREAM_NKT * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.y, b = blockIdx.z, g = lane >> 4;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  bf16_t* ob = reinterpret_cast<bf16_t*>(a.o) + b * a.o_bs + h * D;
  const int nqt = (a.Lq + 15) >> 4;
  const int qt_beg = blockIdx.x * qt_per_block, qt_end = min(nqt, qt_beg + qt_per_block);
  const float sc2 = a.scale * 1.4426950408889634f;
  bf16x8_t qf[STREAM_MAXQ];
  float m[STREAM_MAXQ], l[STREAM_MAXQ];
  f32x4_t o0[STREAM_MAXQ], o1[STREAM_MAXQ];
#pragma unroll
  for (int j = 0; j < STREAM_MAXQ; ++j) {
    const int qt = qt_beg + wave + 4 * j;
    qf[j] = load_frag_global(qb, a.q_rs, qt * 16 + (lane & 15), qt < qt_end ? a.Lq : 0, lane);
    m[j] = -3e38f; l[j] = 0.f;
    o0[j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; o1[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < a.Lk; k0 += STREAM_NKT * 16) {
    __syncthreads();                                               // the previous chunk's images are consumed
    stage_rows2<256>(sK, kb, a.k_rs, sV, vb, a.v_rs, k0, STREAM_NKT * 16, a.Lk, tid);
    __syncthreads();
    const int nkt = min(STREAM_NKT, (((a.Lk - k0 + 15) >> 4) + 1) & ~1);        // even; rows past Lk are zero in LDS and masked below
    const bool tail = k0 + nkt * 16 > a.Lk;
#pragma unroll
    for (int j = 0; j < STREAM_MAXQ; ++j) {
      if (qt_beg + wave + 4 * j >= qt_end) continue;               // wave-uniform
      float mn = m[j];
      for (int kt = 0; kt < nkt; ++kt) {
        const f32x4_t sv = MFMA(frag_n(sK, kt * 16, lane), qf[j], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
        if (tail && k0 + kt * 16 + 16 > a.Lk) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + kt * 16 + g * 4 + r < a.Lk) mn = fmaxf(mn, sv[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) mn = fmaxf(mn, sv[r]);
        }
      }
      mn = group_max4(mn);
      const float alpha = __builtin_amdgcn_exp2f((m[j] - mn) * sc2), mc = mn * sc2;
      m[j] = mn;
      float lj = l[j] * alpha;
      f32x4_t a0 = o0[j] * alpha, a1 = o1[j] * alpha;
      for (int kb2 = 0; kb2 < nkt / 2; ++kb2) {
        f32x4_t pp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kt = 2 * kb2 + t;
          const f32x4_t sv = MFMA(frag_n(sK, kt * 16, lane), qf[j], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pp[t][r] = __builtin_amdgcn_exp2f(fmaf(sv[r], sc2, -mc));
            if (tail && k0 + kt * 16 + g * 4 + r >= a.Lk) pp[t][r] = 0.f;
            lj += pp[t][r];
          }
        }
        const bf16x8_t ph = pack8(pp[0], pp[1]);                   // hi + lo parts of P: see mfma_fwd_kernel
        f32x4_t r0, r1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { r0[r] = pp[0][r] - (float)ph[r]; r1[r] = pp[1][r] - (float)ph[4 + r]; }
        const bf16x8_t pl = pack8(r0, r1);
        const bf16x8_t vt0 = frag_t(sV, kb2 * 32, kb2 * 32 + 16, 0, lane), vt1 = frag_t(sV, kb2 * 32, kb2 * 32 + 16, 16, lane);
        a0 = MFMA(vt0, ph, a0); a0 = MFMA(vt0, pl, a0);
        a1 = MFMA(vt1, ph, a1); a1 = MFMA(vt1, pl, a1);
      }
      l[j] = lj; o0[j] = a0; o1[j] = a1;
    }
  }
#pragma unroll
  for (int j = 0; j < STREAM_MAXQ; ++j) {
    const int qt = qt_beg + wave + 4 * j, q = qt * 16 + (lane & 15);
    if (qt >= qt_end) continue;
    const float lt = group_sum4(l[j]);
    if (q < a.Lq) {
      const float inv = 1.f / lt;
      store8(ob + (int64_t)q * a.o_rs, g, o0[j] * inv, o1[j] * inv);
      if (a.lse && g == 0) a.lse[((int64_t)b * a.H + h) * a.Lq + q] = m[j] * a.scale + __logf(lt);
    }
  }
}

__global__ __launch_bounds__(256) void mfma_bwd_dq_stream_kernel(const AttnArgs a, float* __restrict__ delta, int qt_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[STREAM_NKT * 1024], sV[STREAM_NKT * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.y, b = blockIdx.z, g = lane >> 4;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D;
  const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + h * D;
  const bf16_t* ob = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_bs + h * D;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + b * a.q_bs + h * D;
  const int nqt = (a.Lq + 15) >> 4;
  const int qt_beg = blockIdx.x * qt_per_block, qt_end = min(nqt, qt_beg + qt_per_block);
  const int64_t bh = ((int64_t)b * a.H + h) * a.Lq;
  bf16x8_t qf[STREAM_MAXQ], gf[STREAM_MAXQ];
  float lse[STREAM_MAXQ], dl[STREAM_MAXQ];
  f32x4_t dq0[STREAM_MAXQ], dq1[STREAM_MAXQ];
#pragma unroll
  for (int j = 0; j < STREAM_MAXQ; ++j) {
    const int qt = qt_beg + wave + 4 * j, q = qt * 16 + (lane & 15);
    const int Lq = qt < qt_end ? a.Lq : 0;
    const bool vq = q < Lq;
    qf[j] = load_frag_global(qb, a.q_rs, q, Lq, lane);
    gf[j] = load_frag_global(gb, a.o_rs, q, Lq, lane);
    const bf16x8_t of = load_frag_global(ob, a.o_rs, q, Lq, lane);
    lse[j] = vq ? a.lse[bh + q] : 1e30f;                           // exp(s - 1e30) = 0 masks the padded queries
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += (float)gf[j][e] * (float)of[e];
    d = group_sum4(d);
    if (vq && g == 0) delta[bh + q] = d;
    dl[j] = d;
    dq0[j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dq1[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < a.Lk; k0 += STREAM_NKT * 16) {
    __syncthreads();
    stage_rows2<256>(sK, kb, a.k_rs, sV, vb, a.v_rs, k0, STREAM_NKT * 16, a.Lk, tid);
    __syncthreads();
    const int nkt = min(STREAM_NKT, (((a.Lk - k0 + 15) >> 4) + 1) & ~1);
#pragma unroll
    for (int j = 0; j < STREAM_MAXQ; ++j) {
      if (qt_beg + wave + 4 * j >= qt_end) continue;               // wave-uniform
      f32x4_t a0 = dq0[j], a1 = dq1[j];
      for (int kb2 = 0; kb2 < nkt / 2; ++kb2) {
        f32x4_t ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kt = 2 * kb2 + t;
          const f32x4_t sv = MFMA(frag_n(sK, kt * 16, lane), qf[j], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
          const f32x4_t dp = MFMA(frag_n(sV, kt * 16, lane), gf[j], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = k0 + kt * 16 + g * 4 + r < a.Lk;
            const float p = ok ? __expf(sv[r] * a.scale - lse[j]) : 0.f;
            ds[t][r] = p * (dp[r] - dl[j]) * a.scale;
          }
        }
        const bf16x8_t dsf = pack8(ds[0], ds[1]);
        a0 = MFMA(frag_t(sK, kb2 * 32, kb2 * 32 + 16, 0, lane), dsf, a0);
        a1 = MFMA(frag_t(sK, kb2 * 32, kb2 * 32 + 16, 16, lane), dsf, a1);
      }
      dq0[j] = a0; dq1[j] = a1;
    }
  }
#pragma unroll
  for (int j = 0; j < STREAM_MAXQ; ++j) {
    const int qt = qt_beg + wave + 4 * j, q = qt * 16 + (lane & 15);
    if (qt >= qt_end || q >= a.Lq) continue;
    store8(dqb + (int64_t)q * a.q_rs, g, dq0[j], dq1[j]);
  }
}

// dK / dV of the 14 key tiles starting at tile blockIdx.x * 14: wave w owns tiles w, w + 4, w + 8, w + 12 of the range
// and walks ALL queries, staged chunk by chunk (Q and dO images, LSE, delta) -- no cross-wave or cross-workgroup sums.
__global__ __launch_bounds__(256) void mfma_bwd_dkv_long_kernel(const AttnArgs a, const float* __restrict__ delta, int q_chunk) {
  constexpr int NKT = 14, KW = 4, TPW = 4;
  __shared__ __attribute__((aligned(16))) unsigned char sQG[2 * QR_MAX * 64];
  __shared__ __attribute__((aligned(16))) float sL[QR_MAX], sDl[QR_MAX];
  unsigned char* sQ = sQG;
  unsigned char* sG = sQG + QR_MAX * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.y, b = blockIdx.z, g = lane >> 4;
  const int kt0 = blockIdx.x * NKT;
  const int64_t bh = ((int64_t)b * a.H + h) * a.Lq;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_bs + h * D;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_bs + h * D;
  bf16x8_t kf[TPW], vf[TPW];
  f32x4_t dk[TPW][2], dv[TPW][2];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int lt = wave + i * KW;                                  // tile inside the range
    const int key = (kt0 + lt) * 16 + (lane & 15);
    kf[i] = load_frag_global(kb, a.k_rs, key, (lt < NKT) ? a.Lk : 0, lane);
    vf[i] = load_frag_global(vb, a.v_rs, key, (lt < NKT) ? a.Lk : 0, lane);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { dk[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  }
  for (int q0 = 0; q0 < a.Lq; q0 += q_chunk) {
    const int q1 = min(a.Lq, q0 + q_chunk);
    const int nrows = ((q1 - q0 + 31) >> 5) << 5;
    __syncthreads();                                               // the previous chunk's images are consumed
    stage_rows2<256>(sQ, reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * D, a.q_rs, sG, reinterpret_cast<const bf16_t*>(a.d_o) + b * a.o_bs + h * D, a.o_rs, q0, nrows, q1, tid);
    for (int i = tid; i < nrows; i += 256) {
      const bool ok = q0 + i < q1;
      sL[i] = ok ? a.lse[bh + q0 + i] : 1e30f;                     // exp(s - 1e30) = 0 masks the padded queries
      sDl[i] = ok ? delta[bh + q0 + i] : 0.f;
    }
    __syncthreads();
    for (int r0 = 0; r0 < nrows; r0 += 32) {
      const bf16x8_t qn0 = frag_n(sQ, r0, lane), qn1 = frag_n(sQ, r0 + 16, lane);
      const bf16x8_t gn0 = frag_n(sG, r0, lane), gn1 = frag_n(sG, r0 + 16, lane);
      const bf16x8_t qt0 = frag_t(sQ, r0, r0 + 16, 0, lane), qt1 = frag_t(sQ, r0, r0 + 16, 16, lane);
      const bf16x8_t gt0 = frag_t(sG, r0, r0 + 16, 0, lane), gt1 = frag_t(sG, r0, r0 + 16, 16, lane);
      const float4 l0 = *reinterpret_cast<const float4*>(sL + r0 + g * 4), l1 = *reinterpret_cast<const float4*>(sL + r0 + 16 + g * 4);
      const float4 d0 = *reinterpret_cast<const float4*>(sDl + r0 + g * 4), d1 = *reinterpret_cast<const float4*>(sDl + r0 + 16 + g * 4);
      const float lse[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
      const float dl[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int lt = wave + i * KW;
        if (lt >= NKT || (kt0 + lt) * 16 >= a.Lk) continue;       // wave-uniform
        const bool kvalid = (kt0 + lt) * 16 + (lane & 15) < a.Lk;
        const f32x4_t s0 = MFMA(qn0, kf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f})), s1 = MFMA(qn1, kf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
        const f32x4_t p0 = MFMA(gn0, vf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f})), p1 = MFMA(gn1, vf[i], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
        f32x4_t pr[2], ds[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e0 = kvalid ? __expf(s0[r] * a.scale - lse[r]) : 0.f, e1 = kvalid ? __expf(s1[r] * a.scale - lse[4 + r]) : 0.f;
          pr[0][r] = e0; pr[1][r] = e1;
          ds[0][r] = e0 * (p0[r] - dl[r]) * a.scale; ds[1][r] = e1 * (p1[r] - dl[4 + r]) * a.scale;
        }
        const bf16x8_t pf = pack8(pr[0], pr[1]), dsf = pack8(ds[0], ds[1]);
        dv[i][0] = MFMA(gt0, pf, dv[i][0]); dv[i][1] = MFMA(gt1, pf, dv[i][1]);
        dk[i][0] = MFMA(qt0, dsf, dk[i][0]); dk[i][1] = MFMA(qt1, dsf, dk[i][1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int lt = wave + i * KW, key = (kt0 + lt) * 16 + (lane & 15);
    if (lt >= NKT || key >= a.Lk) continue;
    store8(reinterpret_cast<bf16_t*>(a.dk) + b * a.k_bs + (int64_t)key * a.k_rs + h * D, g, dk[i][0], dk[i][1]);
    store8(reinterpret_cast<bf16_t*>(a.dv) + b * a.v_bs + (int64_t)key * a.v_rs + h * D, g, dv[i][0], dv[i][1]);
  }
}

inline int nkt_for(int Lk) { return Lk <= 32 ? 2 : (Lk <= 64 ? 4 : (Lk <= 128 ? 8 : 14)); }

// query tiles per workgroup: whole (b, h) in one workgroup when small, else ~32 tiles, but keep the grid >= ~1024
inline int qt_per_block_for(const AttnArgs& a) {
  const int nqt = (a.Lq + 15) / 16;
  int per = nqt < 32 ? nqt : 32;
  while (per > 4 && (int64_t)a.B * a.H * ((nqt + per - 1) / per) < 1024) per = (per + 1) / 2;
  return per < 4 ? (nqt < 4 ? nqt : 4) : per;
}

}  // namespace

bool lmv_attn_mfma_long_supported(const AttnArgs& a) { return a.Lk > 224 && a.Lq > 16; }

// query tiles per workgroup of the streaming kernels: <= 4 waves x STREAM_MAXQ, fewer while the grid is short of ~1024
static int stream_per(const AttnArgs& a) {
  const int nqt = (a.Lq + 15) / 16;
  int per = 4 * STREAM_MAXQ;
  while (per > 4 && (int64_t)a.B * a.H * ((nqt + per - 1) / per) < 1024) per /= 2;
  return per;
}

static int long_geometry(const AttnArgs& a, int* nkt, int* per, int* lds) {
  *nkt = ((a.Lk + 31) / 32) * 2;                                   // even number of 16-key tiles
  *lds = 2 * *nkt * 1024;
  const int nqt = (a.Lq + 15) / 16, per0 = qt_per_block_for(a), nblk = (nqt + per0 - 1) / per0;
  *per = (nqt + nblk - 1) / nblk;                                  // balanced query-tile ranges
  // > 64 KiB of dynamic LDS needs an explicit opt-in, once per device; idempotent, so a race through the first call is benign
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    const int max_lds = 2 * LONG_MAX_NKT * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_fwd_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_bwd_dq_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
      LMV_FAIL(LMV_ERR_LAUNCH, "attn_mfma_long: cannot reserve %d bytes of LDS", max_lds);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  return LMV_OK;
}

int lmv_attn_mfma_long_fwd(const AttnArgs& a, hipStream_t st) {
  if (a.Lk > LONG_MAX_NKT * 16) {                                  // K / V streamed through LDS
    const int per = stream_per(a), nqt = (a.Lq + 15) / 16;
    hipLaunchKernelGGL(mfma_fwd_stream_kernel, dim3((nqt + per - 1) / per, a.H, a.B), dim3(256), 0, st, a, per);
    LMV_CHECK_LAUNCH("attn_mfma_stream_fwd");
    return LMV_OK;
  }
  int nkt, per, lds;
  if (int rc = long_geometry(a, &nkt, &per, &lds)) return rc;
  const int nqt = (a.Lq + 15) / 16;
  hipLaunchKernelGGL(mfma_fwd_long_kernel, dim3((nqt + per - 1) / per, a.H, a.B), dim3(256), lds, st, a, per, nkt);
  LMV_CHECK_LAUNCH("attn_mfma_long_fwd");
  return LMV_OK;
}

// delta: B*H*Lq floats of scratch (written by the dQ kernel, read by the dK / dV kernel)
int lmv_attn_mfma_long_bwd(const AttnArgs& a, float* delta, hipStream_t st) {
  const int nqt = (a.Lq + 15) / 16;
  if (a.Lk > LONG_MAX_NKT * 16) {
    const int per = stream_per(a);
    hipLaunchKernelGGL(mfma_bwd_dq_stream_kernel, dim3((nqt + per - 1) / per, a.H, a.B), dim3(256), 0, st, a, delta, per);
  } else {
    int nkt, per, lds;
    if (int rc = long_geometry(a, &nkt, &per, &lds)) return rc;
    hipLaunchKernelGGL(mfma_bwd_dq_long_kernel, dim3((nqt + per - 1) / per, a.H, a.B), dim3(256), lds, st, a, delta, per, nkt);
  }
  const int nchunks = (a.Lq + QR_MAX - 1) / QR_MAX;
  const int q_chunk = (((a.Lq + nchunks - 1) / nchunks) + 31) / 32 * 32;     // balanced chunks, multiples of 32, <= QR_MAX
  const int nranges = ((a.Lk + 15) / 16 + 13) / 14;
  hipLaunchKernelGGL(mfma_bwd_dkv_long_kernel, dim3(nranges, a.H, a.B), dim3(256), 0, st, a, (const float*)delta, q_chunk);
  LMV_CHECK_LAUNCH("attn_mfma_long_bwd");
  return LMV_OK;
}

bool lmv_attn_mfma_supported(const AttnArgs& a) { return a.Lk <= 224; }
bool lmv_attn_mfma_fewq_supported(const AttnArgs& a) { return a.Lq <= 16 && a.Lk > 224; }
int lmv_attn_mfma_fewq_nsplit(const AttnArgs& a) { return (a.Lk + RK - 1) / RK; }

// forward partials [B][H][nsplit][16][34] (max, sum, unnormalised O); the caller runs fwd_fewq_combine_kernel on them
int lmv_attn_mfma_fewq_fwd(const AttnArgs& a, float* part, hipStream_t st) {
  const int nsplit = lmv_attn_mfma_fewq_nsplit(a);
  hipLaunchKernelGGL(mfma_fwd_split_kernel, dim3((nsplit + 3) / 4, a.H, a.B), dim3(256), 0, st, a, part, nsplit);
  LMV_CHECK_LAUNCH("attn_mfma_fewq_fwd");
  return LMV_OK;
}

size_t lmv_attn_mfma_fewq_bwd_acc_bytes(const AttnArgs& a) {
  const int nr = (a.Lk + RKB - 1) / RKB, nblk = (nr + 3) / 4;
  return (size_t)nblk * a.B * a.H * 16 * D * sizeof(float);
}

// acc: lmv_attn_mfma_fewq_bwd_acc_bytes() of scratch: one fp32 dQ slab [B][H][16][32] per workgroup of the key split
int lmv_attn_mfma_fewq_bwd(const AttnArgs& a, float* acc, hipStream_t st) {
  const int nr = (a.Lk + RKB - 1) / RKB, nblk = (nr + 3) / 4;
  hipLaunchKernelGGL(mfma_bwd_fewq_kernel, dim3(nblk, a.H, a.B), dim3(256), 0, st, a, acc);
  const unsigned tot = (unsigned)a.B * a.H * a.Lq * (D / 8);
  hipLaunchKernelGGL(scatter_sum1_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, (const float*)acc, (bf16_t*)a.dq, a.q_bs, a.q_rs, a.B, a.H, a.Lq, 16, nblk);
  LMV_CHECK_LAUNCH("attn_mfma_fewq_bwd");
  return LMV_OK;
}

// dK / dV: split the query range so that the grid fills the chip (and fits the LDS images)
static void dkv_split(const AttnArgs& a, int* nsplit, int* qpb) {
  int ns = (a.Lq + QR_MAX - 1) / QR_MAX;
  while ((int64_t)a.B * a.H * ns < 1024 && a.Lq / (ns + 1) >= 128) ++ns;
  int q = (((a.Lq + ns - 1) / ns) + 31) / 32 * 32;
  if (q > QR_MAX) q = QR_MAX;
  *qpb = q;
  *nsplit = (a.Lq + q - 1) / q;
}

size_t lmv_attn_mfma_bwd_acc_bytes(const AttnArgs& a) {
  int nsplit, qpb;
  dkv_split(a, &nsplit, &qpb);
  return 2 * (size_t)nsplit * a.B * a.H * nkt_for(a.Lk) * 16 * D * sizeof(float);
}

int lmv_attn_mfma_fwd(const AttnArgs& a, hipStream_t st) {
  const int per = qt_per_block_for(a), nqt = (a.Lq + 15) / 16;
  dim3 grid((nqt + per - 1) / per, a.H, a.B), block(256);
  const int pv16 = lmv_config().attn_pv16;      // A/B testing
  if (a.Lk == 196 && pv16) hipLaunchKernelGGL((mfma_fwd_kernel<14, 196, true>), grid, block, 0, st, a, per);      // stage-3 self-attention at 224^2
  else if (a.Lk == 16 && pv16) hipLaunchKernelGGL((mfma_fwd_kernel<2, 16, true>), grid, block, 0, st, a, per);    // 16 meta-token keys (DCA x direction, meta self-attention)
  else if (a.Lk == 49 && pv16) hipLaunchKernelGGL((mfma_fwd_kernel<4, 49, true>), grid, block, 0, st, a, per);    // stage 4
  else
  switch (nkt_for(a.Lk)) {
    case 2: hipLaunchKernelGGL((mfma_fwd_kernel<2>), grid, block, 0, st, a, per); break;
    case 4: hipLaunchKernelGGL((mfma_fwd_kernel<4>), grid, block, 0, st, a, per); break;
    case 8: hipLaunchKernelGGL((mfma_fwd_kernel<8>), grid, block, 0, st, a, per); break;
    default: hipLaunchKernelGGL((mfma_fwd_kernel<14>), grid, block, 0, st, a, per); break;
  }
  LMV_CHECK_LAUNCH("attn_mfma_fwd");
  return LMV_OK;
}

// acc: lmv_attn_mfma_bwd_acc_bytes() of fp32 scratch (only touched when the query range is split)
int lmv_attn_mfma_bwd(const AttnArgs& a, float* delta, float* acc, hipStream_t st) {
  const int per = qt_per_block_for(a), nqt = (a.Lq + 15) / 16, nkt = nkt_for(a.Lk);
  const bool fuse_dq = lmv_config().attn_fuse_dq != 0, fused_bwd = lmv_config().attn_fused_bwd != 0;      // A/B testing
  if (fused_bwd && nkt >= 4 && a.Lq > 16 && a.Lq <= nkt * 16) {
    // one workgroup per (b, h): dQ, dK and dV from ONE pass over the scores
    dim3 grid(1, a.H, a.B), block(256);
    const bool v2 = lmv_config().attn_fused_bwd == 2;          // round 5: dQ by query tile, one barrier per block (mfma_bwd_fused_body2)
    if (a.Lk == 196 && v2) hipLaunchKernelGGL((mfma_bwd_fused_kernel<14, 196, 2>), grid, block, 0, st, a);
    else if (a.Lk == 49 && v2) hipLaunchKernelGGL((mfma_bwd_fused_kernel<4, 49, 2>), grid, block, 0, st, a);
    else if (a.Lk == 196) hipLaunchKernelGGL((mfma_bwd_fused_kernel<14, 196>), grid, block, 0, st, a);
    else if (a.Lk == 49) hipLaunchKernelGGL((mfma_bwd_fused_kernel<4, 49>), grid, block, 0, st, a);
    else if (nkt == 4) hipLaunchKernelGGL((mfma_bwd_fused_kernel<4>), grid, block, 0, st, a);
    else if (nkt == 8) hipLaunchKernelGGL((mfma_bwd_fused_kernel<8>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((mfma_bwd_fused_kernel<14>), grid, block, 0, st, a);
    LMV_CHECK_LAUNCH("attn_mfma_bwd_fused");
    return LMV_OK;
  }
  if (nkt == 14 && a.Lq > 16) {
    // 129..224 keys: the run-time-bound dQ loop (86 registers, 5 waves per SIMD) beats the fully unrolled one (169 registers, 2)
    int nk, pr, lds;
    if (int rc = long_geometry(a, &nk, &pr, &lds)) return rc;
    hipLaunchKernelGGL(mfma_bwd_dq_long_kernel, dim3((nqt + pr - 1) / pr, a.H, a.B), dim3(256), lds, st, a, delta, pr, nk);
  } else if (nkt == 2 && fuse_dq) {
    // <= 32 keys: dQ comes out of the dK / dV kernel below
  } else {
    dim3 grid((nqt + per - 1) / per, a.H, a.B), block(256);
    switch (nkt) {
      case 2: hipLaunchKernelGGL((mfma_bwd_dq_kernel<2>), grid, block, 0, st, a, delta, per); break;
      case 4: hipLaunchKernelGGL((mfma_bwd_dq_kernel<4>), grid, block, 0, st, a, delta, per); break;
      case 8: hipLaunchKernelGGL((mfma_bwd_dq_kernel<8>), grid, block, 0, st, a, delta, per); break;
      default: hipLaunchKernelGGL((mfma_bwd_dq_kernel<14>), grid, block, 0, st, a, delta, per); break;
    }
  }
  int nsplit, qpb;
  dkv_split(a, &nsplit, &qpb);
  dim3 grid(nsplit, a.H, a.B), block(256);
  const size_t acc_elems = (size_t)nsplit * a.B * a.H * nkt * 16 * D;
  float* acc_k = acc; float* acc_v = acc + acc_elems;
  if (nsplit > 1) {
#define DKV(N, K) hipLaunchKernelGGL((mfma_bwd_dkv_kernel<N, K, true>), grid, block, 0, st, a, delta, acc_k, acc_v, qpb)
    if (nkt == 2 && fuse_dq && a.Lk == 16) hipLaunchKernelGGL((mfma_bwd_dkv_kernel<2, 1, true, true, 16>), grid, block, 0, st, a, delta, acc_k, acc_v, qpb);
    else if (nkt == 2 && fuse_dq) hipLaunchKernelGGL((mfma_bwd_dkv_kernel<2, 1, true, true>), grid, block, 0, st, a, delta, acc_k, acc_v, qpb);
    else
    switch (nkt) { case 2: DKV(2, 1); break; case 4: DKV(4, 2); break; case 8: DKV(8, 4); break; default: DKV(14, 4); break; }
#undef DKV
    const unsigned n = 2u * (unsigned)a.B * a.H * a.Lk * (D / 4);
    hipLaunchKernelGGL(scatter_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)acc_k, (const float*)acc_v, (bf16_t*)a.dk, (bf16_t*)a.dv,
                       a.k_bs, a.k_rs, a.v_bs, a.v_rs, a.B, a.H, a.Lk, nkt * 16, nsplit);
  } else {
#define DKV(N, K) hipLaunchKernelGGL((mfma_bwd_dkv_kernel<N, K, false>), grid, block, 0, st, a, delta, acc_k, acc_v, qpb)
    if (nkt == 2 && fuse_dq && a.Lk == 16) hipLaunchKernelGGL((mfma_bwd_dkv_kernel<2, 1, false, true, 16>), grid, block, 0, st, a, delta, acc_k, acc_v, qpb);
    else if (nkt == 2 && fuse_dq) hipLaunchKernelGGL((mfma_bwd_dkv_kernel<2, 1, false, true>), grid, block, 0, st, a, delta, acc_k, acc_v, qpb);
    else
    switch (nkt) { case 2: DKV(2, 1); break; case 4: DKV(4, 2); break; case 8: DKV(8, 4); break; default: DKV(14, 4); break; }
#undef DKV
  }
  LMV_CHECK_LAUNCH("attn_mfma_bwd");
  return LMV_OK;
}

// Pair launches (two independent problems with the same B and H): merged when problem 1 is the stage-3 / stage-4 image-token
// self-attention (196 / 49 keys, Lq == Lk) and problem 2 the 16 x 16 meta-token one; 1 = merged, 0 = not applicable
int lmv_attn_mfma_fwd_pair(const AttnArgs& a1, const AttnArgs& a2, hipStream_t st) {
  const int on = lmv_config().attn_pair;      // A/B testing
  if (!on || a1.B != a2.B || a1.H != a2.H || a2.Lk != 16 || a2.Lq != 16 || a1.Lq != a1.Lk || (a1.Lk != 196 && a1.Lk != 49)) return 0;
  const int per1 = qt_per_block_for(a1), nqt1 = (a1.Lq + 15) / 16, nblk1 = (nqt1 + per1 - 1) / per1;
  dim3 grid((nblk1 + 1) * a1.H * a1.B), block(256);
  if (a1.Lk == 196) hipLaunchKernelGGL((mfma_fwd_pair_kernel<14, 196, 2, 16>), grid, block, 0, st, a1, a2, per1, 1, nblk1);
  else hipLaunchKernelGGL((mfma_fwd_pair_kernel<4, 49, 2, 16>), grid, block, 0, st, a1, a2, per1, 1, nblk1);
  return 1;
}

int lmv_attn_mfma_bwd_pair(const AttnArgs& a1, const AttnArgs& a2, hipStream_t st) {
  const int on = lmv_config().attn_pair;
  if (!on || a1.B != a2.B || a1.H != a2.H || a2.Lk != 16 || a2.Lq != 16 || a1.Lq != a1.Lk || (a1.Lk != 196 && a1.Lk != 49)) return 0;
  dim3 grid(2 * a1.H * a1.B), block(256);
  const bool v2 = lmv_config().attn_fused_bwd == 2;
  if (a1.Lk == 196 && v2) hipLaunchKernelGGL((mfma_bwd_pair_kernel<14, 196, 2>), grid, block, 0, st, a1, a2);
  else if (v2) hipLaunchKernelGGL((mfma_bwd_pair_kernel<4, 49, 2>), grid, block, 0, st, a1, a2);
  else if (a1.Lk == 196) hipLaunchKernelGGL((mfma_bwd_pair_kernel<14, 196>), grid, block, 0, st, a1, a2);
  else hipLaunchKernelGGL((mfma_bwd_pair_kernel<4, 49>), grid, block, 0, st, a1, a2);
  return 1;
}
