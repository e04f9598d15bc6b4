// stage_common.h -- device helpers shared by the persistent stage kernels (csrc/sstage.hip, csrc/dstage.hip): MFMA wrappers, packing, the
// fragment-order GEMM unit (weights straight from L2, token operand from LDS), the interleaved GELU, the in-launch flag wait.
#pragma once
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int SS_NT = 7;              // token tiles of 16 per workgroup (112 slots)
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ bf16x8_t as_bf8(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ bf16x8_t as_bf8(const u32x4_t& v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ f32x4_t mfma_bf16(const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4_t mfma_f16(const f16x8_t& a, const f16x8_t& b, const f32x4_t& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ unsigned pack_h2(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
__device__ __forceinline__ u32x4_t pack_bf8(const f32x4_t& a, const f32x4_t& b) {
  return u32x4_t{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
}
// max without the canonicalising v_max x, x that fmaxf costs under IEEE mode: v_med3(a, b, +inf) = max(a, b) for non-NaN inputs
__device__ __forceinline__ float max2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }
__device__ __forceinline__ float max4(const f32x4_t& s) { return max2(max2(s[0], s[1]), max2(s[2], s[3])); }
// reductions over the 4 lane groups of a token (lanes i, i + 16, i + 32, i + 48): v_permlane16_swap / v_permlane32_swap of two copies leave every lane with both
// halves of a pair -- VALU only (the __shfl_xor form is two ds_bpermute round trips through the LDS pipe per reduction)
__device__ __forceinline__ float xsum4(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float xmax4(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = max2(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max2(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

// one lane polls a flag word until it reaches `epoch` (relaxed agent-scope loads + s_sleep); a bounded spin reports through the error word (pinned host memory: system scope)
__device__ __forceinline__ void wait_flag(unsigned* flag, unsigned epoch, unsigned* err, int lane) {
  if (lane == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load((gu32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > SPIN_LIMIT) { __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
  }
}

// ---- placement-independent role assignment of the workgroups of a persistent stage launch ------------------------------------------------------------------------
// The workgroups of a slot (the two halves of an image in sstage; the image-row workgroups + the meta workgroup of an image slot in dstage) wait for each other inside the
// launch, so a slot only makes progress once ALL its workgroups are resident.  HIP promises nothing about dispatch order or workgroup -> XCD placement
// (MI355X_MICROARCH.md, "Workgroup dispatch": contract), so (slot, role) is not derived from blockIdx: a workgroup TAKES A TICKET when it starts.  tickets[8]: one counter
// per XCD (zeroed by the memset in front of the launch), `quota` tickets each; ticket t of counter y is role t % NWG of slot (t / NWG) * 8 + y.  A workgroup first asks the counter of
// the XCD it really runs on (HW_REG_XCC_ID: the workgroups of a slot then share an L2 -- a speed matter only) and moves on to the next counter when that one is used up, so every one
// of the 8 * quota workgroups of the grid gets exactly one ticket whatever the placement was.  Because tickets are handed out in order, the STARTED workgroups always hold a prefix of
// every counter: at most one slot per counter is incomplete at any time, every other started slot is complete and runs to its end without outside help, and the incomplete ones are
// completed by whichever workgroups start next.  Progress therefore needs only that the device can hold 8 (NWG - 1) + 1 workgroups of the launch (checked on the host:
// lmv_*stage_supported / _max_concurrent), not any particular dispatch order; exhausted waits still report through the error word.
// Returns y | (t << 3), or -1 (no ticket left: cannot happen when the grid is 8 * quota workgroups).  Uses the first dword of `smem`; ends with a barrier.
// skew (test switch "stage_ticket_skew", 0 in production): the first counter asked is displaced by a hash of blockIdx -- the roles are then dealt as under a foreign placement
// (slots span XCDs, counters run out unevenly and workgroups fall through to the next ones); results must not change (tests/test_*stage_gpu.py).
__device__ __forceinline__ int stage_ticket(unsigned* tickets, unsigned quota, unsigned char* smem, unsigned skew = 0) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (skew) xcc += skew * ((blockIdx.x * 2654435761u) >> 27);
    int got = -1;
    for (unsigned k = 0; k < 8; ++k) {
      const unsigned y = (xcc + k) & 7;
      const unsigned t = __hip_atomic_fetch_add((gu32*)(tickets + y), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t < quota) { got = (int)(y | (t << 3)); break; }
    }
    *reinterpret_cast<volatile int*>(smem) = got;
  }
  __syncthreads();
  const int r = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem));
  __syncthreads();
  return r;
}

// gelu_poly2 (common.h) on the 8 pre-activations of one D tile pair at once, the four Horner chains interleaved statement by statement: a dependent
// packed op waits a state for its predecessor, and hipcc does not interleave the chains on its own (it emitted them serially with an s_nop each)
__device__ __forceinline__ void gelu4(f32x2_t& a, f32x2_t& b, f32x2_t& c, f32x2_t& d) {
  f32x2_t sa = a * 0.25f, sb = b * 0.25f, sc = c * 0.25f, sd = d * 0.25f;
  sa[0] = __builtin_amdgcn_fmed3f(sa[0], -1.0f, 1.0f); sb[0] = __builtin_amdgcn_fmed3f(sb[0], -1.0f, 1.0f);
  sc[0] = __builtin_amdgcn_fmed3f(sc[0], -1.0f, 1.0f); sd[0] = __builtin_amdgcn_fmed3f(sd[0], -1.0f, 1.0f);
  sa[1] = __builtin_amdgcn_fmed3f(sa[1], -1.0f, 1.0f); sb[1] = __builtin_amdgcn_fmed3f(sb[1], -1.0f, 1.0f);
  sc[1] = __builtin_amdgcn_fmed3f(sc[1], -1.0f, 1.0f); sd[1] = __builtin_amdgcn_fmed3f(sd[1], -1.0f, 1.0f);
  const f32x2_t ua = sa * sa, ub = sb * sb, uc = sc * sc, ud = sd * sd;
  f32x2_t qa = {-1.6300047636032104f, -1.6300047636032104f}, qb = qa, qc = qa, qd = qa;
#define SS_GSTEP(k)                                                                                                   \
  qa = __builtin_elementwise_fma(qa, ua, f32x2_t{k, k}); qb = __builtin_elementwise_fma(qb, ub, f32x2_t{k, k});       \
  qc = __builtin_elementwise_fma(qc, uc, f32x2_t{k, k}); qd = __builtin_elementwise_fma(qd, ud, f32x2_t{k, k})
  SS_GSTEP(7.93373966217041f); SS_GSTEP(-16.877059936523438f); SS_GSTEP(20.921268463134766f); SS_GSTEP(-17.09065055847168f);
  SS_GSTEP(9.8812894821167f); SS_GSTEP(-4.233964920043945f); SS_GSTEP(1.595382571220398f);
#undef SS_GSTEP
  a = a * __builtin_elementwise_fma(sa, qa, f32x2_t{0.5f, 0.5f}); b = b * __builtin_elementwise_fma(sb, qb, f32x2_t{0.5f, 0.5f});
  c = c * __builtin_elementwise_fma(sc, qc, f32x2_t{0.5f, 0.5f}); d = d * __builtin_elementwise_fma(sd, qd, f32x2_t{0.5f, 0.5f});
}

// ---- one GEMM unit: NC output-channel tiles x all 7 token tiles x NKS k-steps ------------------------------------------------------
// ring: the wave's weight fragments, RD - 1 k-steps ahead, straight from L2 (w: wave-uniform byte pointer to fragment 0 of the unit, fragments
// in [k-step][n] order: scalar base + lane * 16 + immediate).  On entry slots 0 .. RD - 2 hold the first k-steps of this unit (ring_fill or the
// previous unit's tail); on exit they hold those of the unit at `wnext`.  TRANS: D[channel][token] = W X^T; else D[token][channel].
__device__ __forceinline__ bf16x8_t ld_frag(const unsigned char* w, int frag, int lane) { return as_bf8(reinterpret_cast<const uint4*>(w + (size_t)frag * 1024)[lane]); }
template <int NC, int RD>
__device__ __forceinline__ void ring_fill(bf16x8_t (&ring)[RD][NC], const unsigned char* w, int lane) {
#pragma unroll
  for (int s = 0; s < RD - 1; ++s)
#pragma unroll
    for (int n = 0; n < NC; ++n) ring[s][n] = ld_frag(w, s * NC + n, lane);
}
template <int NC, int NKS, int RD, bool TRANS, int NTT = SS_NT, int XST = NTT>          // XST: token tiles per k-step of the operand image at `xs` (NTT of them are used, from xs on)
__device__ __forceinline__ void gemm_unit(f32x4_t (&acc)[NTT][NC], bf16x8_t (&ring)[RD][NC], const unsigned char* wcur, const unsigned char* wnext, const unsigned char* xs, int lane) {
  constexpr bool PRE = NKS % RD == 0;          // else: a unit of RD - 1 k-steps, all of them loaded by ring_fill, nothing fetched ahead (wnext unused)
  static_assert(PRE || NKS == RD - 1, "ring phase");
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if constexpr (PRE) {
      const int sp = ks + RD - 1;
#pragma unroll
      for (int n = 0; n < NC; ++n) ring[sp % RD][n] = sp < NKS ? ld_frag(wcur, sp * NC + n, lane) : ld_frag(wnext, (sp - NKS) * NC + n, lane);
    }
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
      const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(xs + ((ks * XST + t) * 64 + lane) * 16);
#pragma unroll
      for (int n = 0; n < NC; ++n) acc[t][n] = TRANS ? mfma_bf16(ring[ks % RD][n], xf, acc[t][n]) : mfma_bf16(xf, ring[ks % RD][n], acc[t][n]);
    }
  }
}

// ---- LayerNorm of the register-resident rows -> bf16 token operand in LDS (fragment order) ------------------------------------------
// The depth-wise 3 x 3 weights as the stage kernels read them: word [c * 9 + tap] = the bf16 pair (w, 0) for even channels, (0, w) for odd ones -- v_dot2c_f32_bf16 of a loaded
// channel pair with it is that channel's tap product; the words replace the fp32 taps in the packed vector (same count), so a lane's 4 channels x 9 taps are 9 float4 loads and no
// conversion in the kernel.
__global__ __launch_bounds__(256) void stage_posw_pack_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int n) {          // w [C][9]
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int c = k / 9;
  out[k] = (c & 1) ? pack_bf2(0.f, w[k]) : pack_bf2(w[k], 0.f);
}

}  // namespace
