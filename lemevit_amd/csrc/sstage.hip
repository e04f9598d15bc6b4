// sstage.hip -- a whole run of "S" blocks (stage 3 of LeMeViT-Base: 18 blocks at 14 x 14 image tokens + 16 meta tokens, C = 384) as ONE
// persistent launch with the token rows resident on chip (round 4; inference form).
//
// Reference math: LeMeBlock.forward_with_x, models/lemevit.py:615-650 (live branch :631-635): x = x + dwconv3x3(x) (:619);
// x = x + attn(norm1(x)); x = x + mlp(norm2(x)); c = c + attn(norm1(c)); c = c + mlp(norm2(c)) with StandardAttention (:185-205: fused qkv,
// softmax(q k^T / sqrt d) v per head, proj), the MLP of :526-530 (exact GELU) and LayerNorm eps 1e-6 (:513,525); x and c share every weight
// but attend separately.  The per-launch schedule (csrc/block.hip) runs a block as ~5 dependent launches whose qkv / attention output /
// hidden tensors each make an HBM round trip; a forward pass of the stage is 90 launches at 0.185 of the MFMA peak.
//
// Decomposition.  A batch of 128 images has 256 half-images for 256 CUs: TWO workgroups per image, split by token ROWS --
//   half 0: image tokens 0..111 (grid rows 0..7: 7 tiles of 16), half 1: image tokens 112..195 (grid rows 8..13, slots 0..83) + 12 pad
//   slots + the 16 meta tokens (slots 96..111): 7 tiles again -- so no FLOP is duplicated.  What the halves exchange per block, through
//   L2 with write-through (sc1) stores, sc1 loads and one relaxed agent-scope flag each (cdna_hip_programming.md Guideline 16, form R1):
//   the K / V fragments of their rows (attention needs all 196 keys) and the one grid row next to the cut (the 3 x 3 position embedding).
// Inside a workgroup (8 waves, 512 threads, one per CU):
//   * the residual stream lives in REGISTERS, fp32, for the whole stage: wave w owns channels 48 w .. 48 w + 47 of all 112 token slots
//     (84 registers per lane, in the C/D layout of D[channel][token] MFMA tiles);
//   * every Linear is computed channel-split: a wave owns a slice of OUTPUT channels, pulls its weight fragments STRAIGHT from L2 into
//     registers (the weights are pre-packed in MFMA-fragment order by lmv_sstage_pack: one fully coalesced 1 KB load per fragment, no LDS,
//     no barrier inside a GEMM) and reads the token operand -- LayerNorm output, attention output or hidden activations, 112 tokens x K --
//     from LDS, stored in fragment order as well (1 KB per [k-step][token tile], lane-linear ds_read_b128: conflict-free);
//   * outputs are produced TRANSPOSED, D[channel][token]: a lane then holds 4 consecutive channels of one token, and two such tiles are
//     exactly the lane's 8 k-slots of the NEXT contraction's token operand (the k-order inside a 32-channel step is a fixed permutation that
//     the packed weights share) -- q, k, the attention output, the hidden activations and the LayerNorm outputs are all written as ready
//     operand fragments, no transposes anywhere; v is computed with the operands swapped, D[token][channel], which is the V^T operand of P V;
//   * attention per (head, 16-query tile) inside one wave: scores transposed (S^T = K Q^T), two passes over the keys (row maximum, then
//     exp2 / sum / P V with S recomputed: K and V fragments are read once per head and pass, every query tile of the head reuses them),
//     P and V enter the P V product as fp16;
//   * the MLP runs in 6 chunks of 256 hidden channels: fc1 chunk -> GELU -> LDS (57 KB) -> fc2 partial sums accumulate ON the residual registers.
// Measured crux (tools/native/sstage_probe.hip): the GEMM stream of all 18 blocks in this form runs in 1.5 ms on 256 CUs (1.2 PFLOP/s;
// 1.2 ms on 8 CUs: the difference is the L2 -> CU weight stream at ~10.6 TB/s), against ~3 ms of Linear launches in the per-launch schedule.
#include <atomic>
#include "stage_common.h"

namespace {

constexpr int SS_G = 14, SS_NIMG = 196, SS_M = 16;
constexpr int SS_NCHUNK = 6;          // the MLP runs in 6 chunks of 32 NW hidden channels

// The kernel is written for NW waves of 48 channels each: C = 48 NW, 1.5 NW heads of 32, hidden 192 NW.  NW = 8: stage 3 of LeMeViT-Base / Small-v2
// (C = 384, 12 heads, hidden 1536; one 512-thread workgroup per CU).  NW = 4: stage 3 of LeMeViT-Tiny (C = 192, 6 heads, hidden 768; 256-thread workgroups,
// two per CU).  Everything below that says "wave w owns channels 48 w .." scales with it.
//
// Packed weights of a block, in 1 KB fragments (64 lanes x 16 B).  Fragment (row0, ks) of a weight W [N][K]: lane (g = lane >> 4, i = lane & 15)
// holds W[row0 + i][32 ks + 16 (j >> 2) + 4 g + (j & 3)], j = 0..7 -- the A operand of v_mfma_f32_16x16x32_bf16 under the k-slot permutation
// every token operand of this kernel is written in.
template <int NW> struct SG {
  static constexpr int C = 48 * NW, NH = C / 32, HID = 4 * C;
  static constexpr int KS = C / 32;          // k-steps of 32 channels
  static constexpr int KSC = NW;             // k-steps of fc2 per MLP chunk (32 NW hidden channels)
  static constexpr int PRD = KS % 4 == 0 ? 4 : 3;      // ring depth of the proj GEMM (KS must be a multiple)
  static constexpr int WS_KV = 0;                              // [unit u = 2 h + (0: k, 1: v)][ks][n 2]: rows (1 + (u & 1)) C + 32 (u >> 1) + 16 n of qkv.weight
  static constexpr int WS_Q = 2 * NH * KS * 2;                 // [head h][ks][n 2]: rows 32 h + 16 n
  static constexpr int WS_PROJ = WS_Q + NH * KS * 2;           // [wave w][ks][n 3]: rows 48 w + 16 n of proj.weight
  static constexpr int WS_FC1 = WS_PROJ + NW * KS * 3;         // [chunk c][wave w][ks][n 2]: rows 32 NW c + 32 w + 16 n of mlp.0.weight
  static constexpr int WS_FC2 = WS_FC1 + SS_NCHUNK * NW * KS * 2;      // [chunk c][wave w][ksl KSC][n 3]: rows 48 w + 16 n, columns of k-step KSC c + ksl of mlp.3.weight
  static constexpr int WS_FRAGS = WS_FC2 + SS_NCHUNK * NW * KSC * 3;
  // packed fp32 vectors of a block (reference layouts, concatenated)
  static constexpr int V_N1W = 0, V_N1B = C, V_QKVB = 2 * C, V_PROJB = 5 * C, V_N2W = 6 * C, V_N2B = 7 * C, V_FC1B = 8 * C, V_FC2B = 12 * C, V_POSW = 13 * C,
                       V_POSB = 22 * C, V_FLOATS = 23 * C;
  // LDS
  static constexpr int L_XN = 0, L_XN_BYTES = KS * SS_NT * 1024;          // token operand of qkv / fc1 (LayerNorm output) and of proj (attention output)
  static constexpr int L_H = L_XN_BYTES, L_H_BYTES = KSC * SS_NT * 1024;   // hidden chunk (token operand of fc2); second-head q fragments during attention
  static constexpr int L_STAT = L_H + L_H_BYTES, L_STAT_BYTES = NW * 112 * 8;   // LayerNorm partial sums [wave][slot] float2
  static constexpr int L_TOTAL = L_STAT + L_STAT_BYTES;                    // NW = 8: 150 528 B, NW = 4: 75 264 B
  // workspace
  static constexpr size_t KBUF_IMG = (size_t)NH * 14 * 1024;      // [head][14 key tiles: image 0..12, meta 13][1 KB]
  static constexpr size_t VBUF_IMG = (size_t)NH * 8 * 2 * 1024;   // [head][8 key-tile pairs: half 0 p0..3, half 1 p0..3][2 d-tiles][1 KB], fp16
  static constexpr size_t HALO_IMG = (size_t)2 * 14 * C * 2;      // [half][14 tokens][C] bf16
  static constexpr size_t PARK_IMG = (size_t)2 * NW * 21 * 1024;  // [half][wave][21 residual tiles][1 KB]: the fp32 residual registers, parked in L2 while k / v / q / attention run
};
// dwconv staging: per wave [10 grid rows][16 columns][48 channels] bf16 (over L_XN | L_H).  An entry is 96 bytes of channels in a 104-byte slot (round 5): the 16 lanes of a
// ds_read2_b64 lane group read 8 bytes each of 16 CONSECUTIVE entries, and with the 24-dword stride of packed entries 24 p mod 32 takes 4 values -- a 4-way bank conflict on every
// tap read, which is what the phase was made of (~1000 cycles per token tile for ~62 instructions; tools/native/icache_probe.hip ruled the instruction cache out).  26 p mod 32 runs
// through all 16 even banks: conflict-free.  (104 is a multiple of 8 only: every access to the image is 8 bytes wide.)
#ifndef SS_STG_ROW
#define SS_STG_ROW 104
#endif
constexpr int STG_ROW = SS_STG_ROW, STG_WAVE = 160 * STG_ROW;
static_assert(8 * STG_WAVE <= SG<8>::L_STAT && 4 * STG_WAVE <= SG<4>::L_STAT, "staging overlaps the statistics");
// the geometry of SG<NW> under the names the code uses
#define SS_GEO(NW)                                                                                                                                     \
  using G_ = SG<NW>;                                                                                                                                   \
  constexpr int SS_C = G_::C, SS_NH = G_::NH, SS_HID = G_::HID, SS_KS = G_::KS, SS_KSC = G_::KSC, SS_PRD = G_::PRD;                                    \
  constexpr int WS_KV = G_::WS_KV, WS_Q = G_::WS_Q, WS_PROJ = G_::WS_PROJ, WS_FC1 = G_::WS_FC1, WS_FC2 = G_::WS_FC2, WS_FRAGS = G_::WS_FRAGS;          \
  constexpr int V_N1W = G_::V_N1W, V_N1B = G_::V_N1B, V_QKVB = G_::V_QKVB, V_PROJB = G_::V_PROJB, V_N2W = G_::V_N2W, V_N2B = G_::V_N2B,                \
                V_FC1B = G_::V_FC1B, V_FC2B = G_::V_FC2B, V_POSW = G_::V_POSW, V_POSB = G_::V_POSB, V_FLOATS = G_::V_FLOATS;                           \
  constexpr int L_XN = G_::L_XN, L_H = G_::L_H, L_STAT = G_::L_STAT, L_TOTAL = G_::L_TOTAL;                                                            \
  constexpr size_t KBUF_IMG = G_::KBUF_IMG, VBUF_IMG = G_::VBUF_IMG, HALO_IMG = G_::HALO_IMG, PARK_IMG = G_::PARK_IMG;                                 \
  (void)SS_C; (void)SS_NH; (void)SS_HID; (void)SS_KS; (void)SS_KSC; (void)SS_PRD; (void)WS_KV; (void)WS_Q; (void)WS_PROJ; (void)WS_FC1; (void)WS_FC2;  \
  (void)WS_FRAGS; (void)V_N1W; (void)V_N1B; (void)V_QKVB; (void)V_PROJB; (void)V_N2W; (void)V_N2B; (void)V_FC1B; (void)V_FC2B; (void)V_POSW;           \
  (void)V_POSB; (void)V_FLOATS; (void)L_XN; (void)L_H; (void)L_STAT; (void)L_TOTAL; (void)KBUF_IMG; (void)VBUF_IMG; (void)HALO_IMG; (void)PARK_IMG;
constexpr int SS_NSTAMP = 24;
#ifndef SS_LN1_VAR
#define SS_LN1_VAR 0      // LayerNorm statistics: 0 per-wave centred sums + parallel-variance combine; 1 one-pass E[x^2] - mean^2 (rounds 1 - 4; A/B builds); 2 the new partials through the old formula
#endif
#ifndef SS_LN2_VAR
#define SS_LN2_VAR 0
#endif
// The combine of the per-wave LayerNorm partials must stay SCALAR code.  Left alone, hipcc (ROCm 7.2) SLP-packs the arithmetic of two token tiles into v_pk_add / v_pk_fma_f32
// (tile t + 1 in the high lanes), and that build is NOT run-to-run reproducible on gfx950 when two workgroups share a CU (the 4-wave instance at B = 256: ~3 % of the K / V words of the
// ODD token tiles differ between two launches of the same inputs by an ulp of the LayerNorm output; one workgroup per CU, or the 8-wave instance: bit-identical).  Everything that could
// be checked by hand is consistent -- the waitcnt counts, the splat constants (s[34:35] = (eps, eps)), op_sel / op_sel_hi on the SGPR-pair sources (tools/native/pk_sgpr_probe.hip: the
// hardware honours them), no permlane or DS-store hazard (variants with shuffles / nops / drained stores behave the same) -- and the scalar build of the same source is bit-stable
// (tools/ss_repeat.py; tests/test_sstage_gpu.py::test_sstage_vs_per_launch_schedule_full_size is the regression test that caught it).  Opaque copies per tile keep the SLP
// vectoriser off these values (tools/experiments/sstage_ln_packed.patch rebuilds the packed form); tests/test_isa_cpu.py fails the build if a packed fp32 op shows up in the statistics phases.
#define SS_LN_OPAQUE2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#ifndef SS_DBG_LN_BARRIER
#define SS_DBG_LN_BARRIER 0
#endif
#ifndef SS_DBG_NO_DW
#define SS_DBG_NO_DW 0
#endif
#ifndef SS_PARK
#define SS_PARK 1      // park the residual registers in L2 while k / v / q / attention run
#endif
#ifndef SS_DWFENCE
#define SS_DWFENCE 1
#endif
#ifndef SS_QFENCE
#define SS_QFENCE 2
#endif
#ifndef SS_LDAUX
#define SS_LDAUX 16      // sc1: the K / V fragments another workgroup wrote inside this launch (Guideline 16, R1)
#endif

struct SsArgs {
  const bf16_t* x_in; const bf16_t* c_in; bf16_t* x_out; bf16_t* c_out;
  const uint4* wpk; const float* vec;
  unsigned char* kbuf; unsigned char* vbuf; unsigned char* halo; unsigned char* park; unsigned* flags; unsigned* err;     // flags: [2 B] kv | [2 B] halo | [8] tickets; err: the sticky error word (lmv_stage_error_count)
  unsigned* tickets; unsigned quota, skew;      // stage_ticket (stage_common.h): 8 counters of `quota` = 2 ceil(B / 8) tickets
  int B, nblocks; float eps;
  unsigned long long* timing; int timing_block;      // optional (NULL): s_memtime stamps [workgroup][wave][SS_NSTAMP] of one block
};

#define SS_STAMP(k)                                                                                              \
  do {                                                                                                           \
    if (a.timing && blk == a.timing_block && lane == 0) a.timing[((size_t)blockIdx.x * NW + wave) * SS_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

template <int NW, int VAR = 0>
__device__ __forceinline__ void layer_norm_to_lds(const f32x4_t (&R)[SS_NT][3], const float* gam, const float* bet, float eps, unsigned char* smem, int wave, int lane) {
  __builtin_amdgcn_sched_barrier(0); asm volatile("; LN_FENCE_BEGIN" ::: "memory");          // (tests/test_isa_cpu.py: no packed fp32 arithmetic between the markers: the statistics and their combine)
  SS_GEO(NW)
  const int g = lane >> 4, li = lane & 15;
  float2* stat = reinterpret_cast<float2*>(smem + L_STAT);
#if SS_DBG_LN_BARRIER
  __syncthreads();
#endif
  // Statistics without cancellation (the one-pass E[x^2] - mean^2 loses every digit of a row with a large mean and a small variance): a wave takes the
  // mean and the centred sum of squares of ITS 48 channels of a token in two passes over its registers, the NW partial (mean_w, M2_w) pairs are combined with the parallel-variance
  // formula  M2 = sum M2_w + 48 sum (mean_w - mean)^2  (Chan et al.) -- differences of means, never of squares.  No extra barrier, no value lives across phases.
  if constexpr (VAR == 1) {
#pragma unroll
    for (int t = 0; t < SS_NT; ++t) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1 += R[t][ct][r]; s2 = fmaf(R[t][ct][r], R[t][ct][r], s2); }
      s1 = xsum4(s1); s2 = xsum4(s2);
      if (g == 0) stat[wave * 112 + t * 16 + li] = make_float2(s1, s2);
    }
  } else {
#pragma unroll
    for (int t = 0; t < SS_NT; ++t) {
      float s1 = 0.f;
  #pragma unroll
      for (int ct = 0; ct < 3; ++ct)
  #pragma unroll
        for (int r = 0; r < 4; ++r) s1 += R[t][ct][r];
      const float mw = xsum4(s1) * (1.f / 48.f);
      float m2 = 0.f;
  #pragma unroll
      for (int ct = 0; ct < 3; ++ct)
  #pragma unroll
        for (int r = 0; r < 4; ++r) { const float dlt = R[t][ct][r] - mw; m2 = fmaf(dlt, dlt, m2); }
      m2 = xsum4(m2);
      if (g == 0) stat[wave * 112 + t * 16 + li] = make_float2(mw, m2);
    }
  }
  float4 ga[3], be[3];        // requested ahead of the barrier
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) { ga[ct] = *reinterpret_cast<const float4*>(gam + 48 * wave + 16 * ct + 4 * g); be[ct] = *reinterpret_cast<const float4*>(bet + 48 * wave + 16 * ct + 4 * g); }
  __syncthreads();
  float mean[SS_NT], rstd[SS_NT];
  if constexpr (VAR == 1) {
#pragma unroll
    for (int t = 0; t < SS_NT; ++t) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { const float2 p = stat[w * 112 + t * 16 + li]; s1 += p.x; s2 += p.y; }
      mean[t] = s1 * (1.f / SS_C);
      rstd[t] = rsqrtf(fmaxf(s2 * (1.f / SS_C) - mean[t] * mean[t], 0.f) + eps);
    }
  } else if constexpr (VAR == 2) {
#pragma unroll
    for (int t = 0; t < SS_NT; ++t) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { const float2 p = stat[w * 112 + t * 16 + li]; s1 += 48.f * p.x; s2 += fmaf(48.f * p.x, p.x, p.y); }
      mean[t] = s1 * (1.f / SS_C);
      rstd[t] = rsqrtf(fmaxf(s2 * (1.f / SS_C) - mean[t] * mean[t], 0.f) + eps);
    }
  } else {
#pragma unroll
    for (int t = 0; t < SS_NT; ++t) {
      float2 p[NW];
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { p[w] = stat[w * 112 + t * 16 + li]; sm += p[w].x; sq += p[w].y; }
      SS_LN_OPAQUE2(sm, sq);
      mean[t] = sm * (1.f / NW);
      float dev = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { const float dlt = p[w].x - mean[t]; dev = fmaf(dlt, dlt, dev); }
      rstd[t] = rsqrtf(fmaf(dev, 48.f, sq) * (1.f / SS_C) + eps);
      SS_LN_OPAQUE2(mean[t], rstd[t]);
    }
  }
asm volatile("; LN_FENCE_END" ::: "memory"); __builtin_amdgcn_sched_barrier(0);          // (the normalisation below packs two CHANNELS of a token with VGPR-pair splats: that form has always been in the bit-stable builds)
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    const int T = 3 * wave + ct;
#pragma unroll
    for (int t = 0; t < SS_NT; ++t) {
      const float y0 = fmaf((R[t][ct][0] - mean[t]) * rstd[t], ga[ct].x, be[ct].x), y1 = fmaf((R[t][ct][1] - mean[t]) * rstd[t], ga[ct].y, be[ct].y);
      const float y2 = fmaf((R[t][ct][2] - mean[t]) * rstd[t], ga[ct].z, be[ct].z), y3 = fmaf((R[t][ct][3] - mean[t]) * rstd[t], ga[ct].w, be[ct].w);
      *reinterpret_cast<uint2*>(smem + L_XN + (((T >> 1) * SS_NT + t) * 64 + lane) * 16 + (T & 1) * 8) = make_uint2(pack_bf2(y0, y1), pack_bf2(y2, y3));
    }
  }
  __syncthreads();
}

// ---- attention of one head for the NQ image-query tiles T0 .. T0 + NQ - 1 (keys: the 196 image tokens, 13 key tiles) ------------------------
// Qf: B operands (q scaled by log2 e / sqrt d, bf16); K fragments [head][key tile], V fragments [head][pair slot][d-tile] from the exchange
// buffers.  All 13 K fragments of the head are requested at once and stay in registers for both passes (row maximum; exp2 / sum / P V with S
// recomputed), the V fragments run three steps ahead -- every loop is fully unrolled so that no fragment is ever copied between registers: a
// rotating-register form made hipcc wait for the newest load at the top of every step, which exposed one L2 round trip per key tile.  (The
// residual registers are parked in L2 during this phase: K alone is 52 registers.)  A head runs as two groups of query tiles (4 + 3).
constexpr int L_AO = 0;      // (= SG<NW>::L_XN: the attention output overwrites the LayerNorm output)
template <int T0, int NQ>
__device__ __forceinline__ void attn_image(const bf16x8_t (&Qf)[SS_NT], int h, __amdgpu_buffer_rsrc_t kr, __amdgpu_buffer_rsrc_t vr, unsigned char* smem, int lane) {
  const int g = lane >> 4;
  const int kbase = (h * 14 * 64 + lane) * 16, vbase = (h * 16 * 64 + lane) * 16;
  const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
  u32x4_t K[13], V[7][2];
#pragma unroll
  for (int kt = 0; kt < 13; ++kt) K[kt] = __builtin_amdgcn_raw_buffer_load_b128(kr, kbase + kt * 1024, 0, SS_LDAUX);
#pragma unroll
  for (int s = 0; s < 3; ++s) { V[s][0] = __builtin_amdgcn_raw_buffer_load_b128(vr, vbase + (s * 2) * 1024, 0, SS_LDAUX); V[s][1] = __builtin_amdgcn_raw_buffer_load_b128(vr, vbase + (s * 2 + 1) * 1024, 0, SS_LDAUX); }
  float mx[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) mx[q] = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 13; ++kt) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const f32x4_t s = mfma_bf16(as_bf8(K[kt]), Qf[T0 + q], z4);
      const float m = max4(s);
      mx[q] = max2(mx[q], (kt == 12 && g != 0) ? -INFINITY : m);        // key tile 12: only keys 192..195 (rows 0..3 = lane group 0) exist
    }
    if (kt % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) mx[q] = xmax4(mx[q]);
  // the second pass RECOMPUTES the score tiles: without these opaque copies the compiler recognises the first pass's MFMAs and keeps all 13 x NQ
  // score tiles (208 registers at NQ = 4) alive instead
#pragma unroll
  for (int kt = 0; kt < 13; ++kt) asm volatile("" : "+v"(K[kt]));
  // second pass (VALU-bound; the matrix pipe has room): the row maximum rides the score MFMA as its accumulator input (s - m for free; -inf where a key does not exist), the row
  // sums come out of one more MFMA against a fragment of ones (over the 32 keys of the step and over the lane groups: no adds, no cross-lane step)
  f32x4_t O[NQ][2], L[NQ], negm[NQ];
  const f16x8_t ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
#pragma unroll
  for (int q = 0; q < NQ; ++q) { O[q][0] = z4; O[q][1] = z4; L[q] = z4; negm[q] = f32x4_t{-mx[q], -mx[q], -mx[q], -mx[q]}; }
  // key tiles (ta, tb) of step s and their V pair slot: half-0 pairs (0,1) (2,3) (4,5) (6,-), half-1 pairs (7,8) (9,10) (11,12)
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    if (s + 3 < 7) { V[s + 3][0] = __builtin_amdgcn_raw_buffer_load_b128(vr, vbase + ((s + 3) * 2) * 1024, 0, SS_LDAUX); V[s + 3][1] = __builtin_amdgcn_raw_buffer_load_b128(vr, vbase + ((s + 3) * 2 + 1) * 1024, 0, SS_LDAUX); }
    const int ta = s < 4 ? 2 * s : 2 * s - 1;
    const bool dummy = s == 3;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const f32x4_t sa = mfma_bf16(as_bf8(K[ta]), Qf[T0 + q], negm[q]);
      float p[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(sa[r]);
      if (!dummy) {
        const float nb = (s == 6 && g != 0) ? -INFINITY : -mx[q];      // exp2(-inf) = 0: absent keys
        const f32x4_t sb = mfma_bf16(as_bf8(K[ta + 1]), Qf[T0 + q], s == 6 ? f32x4_t{nb, nb, nb, nb} : negm[q]);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 + r] = __builtin_amdgcn_exp2f(sb[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 + r] = 0.f;
      }
      const u32x4_t pk = {pack_h2(p[0], p[1]), pack_h2(p[2], p[3]), pack_h2(p[4], p[5]), pack_h2(p[6], p[7])};
      const f16x8_t pf = __builtin_bit_cast(f16x8_t, pk);
      O[q][0] = mfma_f16(__builtin_bit_cast(f16x8_t, V[s][0]), pf, O[q][0]);
      O[q][1] = mfma_f16(__builtin_bit_cast(f16x8_t, V[s][1]), pf, O[q][1]);
      L[q] = mfma_f16(ones, pf, L[q]);
      if (q % SS_QFENCE == SS_QFENCE - 1) __builtin_amdgcn_sched_barrier(0);      // unfenced, the scheduler pulls every V request to the top and interleaves all steps and tiles (spills)
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float inv = 1.f / L[q][0];
    const u32x4_t o = pack_bf8(O[q][0] * inv, O[q][1] * inv);
    *reinterpret_cast<u32x4_t*>(smem + L_AO + ((h * SS_NT + T0 + q) * 64 + lane) * 16) = o;
  }
}
// the 16 meta queries of half 1 (token tile 6) against the 16 meta keys (key tile 13, V pair slot 7)
__device__ __forceinline__ void attn_meta(const bf16x8_t& Qf, int h, __amdgpu_buffer_rsrc_t kr, __amdgpu_buffer_rsrc_t vr, unsigned char* smem, int lane) {
  const int kbase = (h * 14 * 64 + lane) * 16, vbase = (h * 16 * 64 + lane) * 16;
  const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
  const u32x4_t kf = __builtin_amdgcn_raw_buffer_load_b128(kr, kbase + 13 * 1024, 0, SS_LDAUX);
  const u32x4_t v0 = __builtin_amdgcn_raw_buffer_load_b128(vr, vbase + 14 * 1024, 0, 16), v1 = __builtin_amdgcn_raw_buffer_load_b128(vr, vbase + 15 * 1024, 0, SS_LDAUX);
  const f32x4_t s = mfma_bf16(as_bf8(kf), Qf, z4);
  const float m = xmax4(max4(s));
  float p[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s[r] - m);
  const float inv = 1.f / xsum4((p[0] + p[1]) + (p[2] + p[3]));
  const u32x4_t pk = {pack_h2(p[0], p[1]), pack_h2(p[2], p[3]), 0u, 0u};
  const f16x8_t pf = __builtin_bit_cast(f16x8_t, pk);
  const f32x4_t o0 = mfma_f16(__builtin_bit_cast(f16x8_t, v0), pf, z4), o1 = mfma_f16(__builtin_bit_cast(f16x8_t, v1), pf, z4);
  *reinterpret_cast<u32x4_t*>(smem + L_AO + ((h * SS_NT + 6) * 64 + lane) * 16) = pack_bf8(o0 * inv, o1 * inv);
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void sstage_kernel(const SsArgs a) {
  SS_GEO(NW)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63, wave0 = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Every phase re-derives its lane / wave quantities from an opaque copy (SS_PHASE): otherwise the compiler hoists ~100 loop-invariant
  // addresses and masks out of the block loop and spills them next to the 84 residual registers.
#define SS_PHASE                                                    \
  __builtin_amdgcn_sched_barrier(0);                                \
  int lane = lane0; asm volatile("" : "+v"(lane));                  \
  int wave = wave0; asm volatile("" : "+s"(wave));                  \
  const int g = lane >> 4, li = lane & 15; (void)g; (void)li; (void)wave;
  // (image, half) by ticket, not by blockIdx: whichever workgroups are resident form complete pairs, under any dispatch order (stage_ticket, stage_common.h); the two halves of an
  // image take consecutive tickets of the counter of the XCD they run on, so they share an L2 wherever the hardware places them (a speed matter only)
  const int tk = stage_ticket(a.tickets, a.quota, smem, a.skew);
  if (tk < 0) return;
  const int xcd = tk & 7, idx = tk >> 3;
  const int img = (idx >> 1) * 8 + xcd, half = idx & 1;
  if (img >= a.B) return;
  unsigned* const kvflag_mine = a.flags + img * 2 + half;
  unsigned* const kvflag_peer = a.flags + img * 2 + (1 - half);
  unsigned* const haloflag_mine = a.flags + 2 * a.B + img * 2 + half;
  unsigned* const haloflag_peer = a.flags + 2 * a.B + img * 2 + (1 - half);
  unsigned* const errflag = a.err;
  const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(a.kbuf + (size_t)img * KBUF_IMG, 0, (int)KBUF_IMG, 0x00020000);
  const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(a.vbuf + (size_t)img * VBUF_IMG, 0, (int)VBUF_IMG, 0x00020000);
  const int nimg_slots = half ? 84 : 112;          // slots that are image tokens
  const int tok0 = half ? 112 : 0;

  // ---- residual rows -> registers (fp32): R[t][ct][r] = token slot 16 t + li, channel 48 wave + 16 ct + 4 g + r ----
  f32x4_t R[SS_NT][3];
  {
  SS_PHASE
#pragma unroll
  for (int t = 0; t < SS_NT; ++t) {
    const int slot = 16 * t + li;
    const bf16_t* src = nullptr;
    if (slot < nimg_slots) src = a.x_in + ((size_t)img * SS_NIMG + tok0 + slot) * SS_C;
    else if (half && slot >= 96) src = a.c_in + ((size_t)img * SS_M + slot - 96) * SS_C;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
      float f[4] = {0.f, 0.f, 0.f, 0.f};
      if (src) ld4(src + 48 * wave + 16 * ct + 4 * g, f);
      R[t][ct] = f32x4_t{f[0], f[1], f[2], f[3]};
    }
  }
  }

#pragma unroll 1
  for (int blk = 0; blk < a.nblocks; ++blk) {
    const unsigned char* const wp = reinterpret_cast<const unsigned char*>(a.wpk) + (size_t)blk * WS_FRAGS * 1024;      // wave-uniform
    const float* const vec = a.vec + (size_t)blk * V_FLOATS;
    const int lane = lane0, wave = wave0;      // (stamps only)

    SS_STAMP(0);
    asm volatile("; PHASE_DWCONV" ::: "memory");
    // ---- x += dwconv3x3(x) + bias on the 14 x 14 grid (models/lemevit.py:619): the wave's 48 channels of its tokens and of the one grid
    //      row across the cut go through a wave-private bf16 staging image; taps are v_dot2c_f32_bf16 on a loaded channel pair, the sum is added to the fp32 residual ----
    {
      SS_PHASE
      // staging image of the wave's 48 channels: [10 grid rows: the row above, <= 8 own rows, the row below][16 columns: zero | x = 0..13 | zero],
      // 96 B per entry.  The row across the cut comes from the peer, the row beyond the grid and the two pad columns are zero-filled: every
      // tap is then an unconditional read at base + an immediate offset, no bounds masks (they were half of this phase's VALU work).
      unsigned char* const stg = smem + wave * STG_WAVE;
      // the 4 x 9 tap weights + bias of the lane's channels, one channel tile ahead (an L2 round trip per tile otherwise); the first tile's are requested HERE, in front of
      // the staging stores and the halo wait (round 5: behind them the first token tile of the phase waited ~2 k cycles for this round trip)
      float4 wq[2][10];
#pragma unroll
      for (int e = 0; e < 9; ++e) wq[0][e] = *reinterpret_cast<const float4*>(vec + V_POSW + (48 * wave + 4 * g) * 9 + 4 * e);
      wq[0][9] = *reinterpret_cast<const float4*>(vec + V_POSB + 48 * wave + 4 * g);
#pragma unroll
      for (int t = 0; t < SS_NT; ++t) {
        const int slot = 16 * t + li;
        if (slot < nimg_slots) {
          const int y = slot / SS_G, x = slot - y * SS_G;
#pragma unroll
          for (int ct = 0; ct < 3; ++ct)
            *reinterpret_cast<uint2*>(stg + ((y + 1) * 16 + x + 1) * STG_ROW + 32 * ct + 8 * g) = make_uint2(pack_bf2(R[t][ct][0], R[t][ct][1]), pack_bf2(R[t][ct][2], R[t][ct][3]));
        }
      }
      {
        const int grow = half ? 7 : 0;           // the grid row beyond the image: zeros
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int p = lane + 64 * it;
          if (p < 36 * 6) {
            const int e = p / 6, q = p - e * 6;
            const int idx = e < 20 ? (e >> 1) * 16 + (e & 1) * 15 : grow * 16 + (e - 20);
            *reinterpret_cast<uint2*>(stg + idx * STG_ROW + 16 * q) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(stg + idx * STG_ROW + 16 * q + 8) = make_uint2(0u, 0u);
          }
        }
      }
      SS_STAMP(13);
      {
        const unsigned char* hsrc;
        if (blk == 0) hsrc = reinterpret_cast<const unsigned char*>(a.x_in + ((size_t)img * SS_NIMG + (half ? 98 : 112)) * SS_C);
        else { wait_flag(haloflag_peer, (unsigned)blk, errflag, lane); hsrc = a.halo + (size_t)img * HALO_IMG + (size_t)(1 - half) * 14 * SS_C * 2; }
        const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hsrc), 0, 14 * SS_C * 2, 0x00020000);
        const int hrow = half ? 0 : 9;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int p = lane + 64 * it;
          if (p < 84) {
            const int tok = p / 6, q = p - tok * 6;
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(hr, tok * SS_C * 2 + (48 * wave + 8 * q) * 2, 0, 16);
            *reinterpret_cast<uint2*>(stg + (hrow * 16 + tok + 1) * STG_ROW + 16 * q) = make_uint2(v[0], v[1]);          // (the image is bf16, as the exchanged rows are)
            *reinterpret_cast<uint2*>(stg + (hrow * 16 + tok + 1) * STG_ROW + 16 * q + 8) = make_uint2(v[2], v[3]);
          }
        }
      }
      SS_STAMP(14);
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        int l2 = lane; asm volatile("" : "+v"(l2));      // per-channel-tile copies: the per-token addresses are recomputed, not kept across the three passes
        const int g = l2 >> 4, li = l2 & 15;
        if (ct + 1 < 3) {
          const int c1 = 48 * wave + 16 * (ct + 1) + 4 * g;
#pragma unroll
          for (int e = 0; e < 9; ++e) wq[(ct + 1) & 1][e] = *reinterpret_cast<const float4*>(vec + V_POSW + c1 * 9 + 4 * e);
          wq[(ct + 1) & 1][9] = *reinterpret_cast<const float4*>(vec + V_POSB + c1);
        }
        // tap weights: bf16 pairs (w, 0) / (0, w), packed by lmv_*stage_pack (stage_common.h): v_dot2c_f32_bf16 of a loaded channel pair with one of them is that channel's tap product,
        // accumulated in fp32 (bf16 taps, as the reference's autocast convolution; the fp16 image of an earlier version overflowed on residual streams beyond 65504)
        unsigned wt[36];
#pragma unroll
        for (int e = 0; e < 9; ++e) { const float4 v = wq[ct & 1][e]; wt[4 * e] = __float_as_uint(v.x); wt[4 * e + 1] = __float_as_uint(v.y); wt[4 * e + 2] = __float_as_uint(v.z); wt[4 * e + 3] = __float_as_uint(v.w); }
        const float4 pb = wq[ct & 1][9];
        // the 9 taps of token tile t + 1 are requested BEFORE the 36 dot products of tile t are issued (round 5: two statically named register sets; with one set the phase sat at ~2x its
        // instruction-issue time -- 9 LDS round trips exposed per tile at two waves per SIMD)
        uint2 f[2][9];
        auto tap_base = [&](int t) -> const unsigned char* {
          const int slot = 16 * t + li;
          const int sv = slot < nimg_slots ? slot : 0, y = sv / SS_G, x = sv - y * SS_G;
          return stg + (y * 16 + x) * STG_ROW + 32 * ct + 8 * g;      // entry of the (-1, -1) neighbour; tap (dy, dx): + ((dy + 1) * 16 + dx + 1) * STG_ROW
        };
        {
          const unsigned char* const tap0 = tap_base(0);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) f[0][tap] = *reinterpret_cast<const uint2*>(tap0 + ((tap / 3) * 16 + tap % 3) * STG_ROW);
        }
#pragma unroll
        for (int t = 0; t < SS_NT; ++t) {
          const int slot = 16 * t + li;
#if SS_DBG_NO_DW
          const bool valid = false;
#else
          const bool valid = slot < nimg_slots;
#endif
          if (t + 1 < SS_NT) {
            const unsigned char* const tap1 = tap_base(t + 1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) f[(t + 1) & 1][tap] = *reinterpret_cast<const uint2*>(tap1 + ((tap / 3) * 16 + tap % 3) * STG_ROW);
          }
          float acc[4] = {pb.x, pb.y, pb.z, pb.w};
          // (the builtin, not inline asm: a DOT result needs wait states before another VALU instruction reads it, and the hazard recognizer only covers instructions it can see); one
          // v_dot2c_f32_bf16 per tap and channel takes the bf16 operand straight from a loaded pair (unpacking costs two more operations per pair)
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) asm volatile("" : "+v"(f[t & 1][tap]));
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            acc[0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, f[t & 1][tap].x), __builtin_bit_cast(bf16x2_t, wt[0 * 9 + tap]), acc[0], false);
            acc[1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, f[t & 1][tap].x), __builtin_bit_cast(bf16x2_t, wt[1 * 9 + tap]), acc[1], false);
            acc[2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, f[t & 1][tap].y), __builtin_bit_cast(bf16x2_t, wt[2 * 9 + tap]), acc[2], false);
            acc[3] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, f[t & 1][tap].y), __builtin_bit_cast(bf16x2_t, wt[3 * 9 + tap]), acc[3], false);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) R[t][ct][r] += valid ? acc[r] : 0.f;
          // one token tile's 9 taps at a time, finished here: otherwise LLVM sinks the FMA chains to the first use of R (the LayerNorm) and
          // keeps all 189 loaded taps alive (in scratch) until then
          asm volatile("" : "+v"(R[t][ct]));
          if (t % SS_DWFENCE == SS_DWFENCE - 1) __builtin_amdgcn_sched_barrier(0);
          if (ct == 0 && t == 0) SS_STAMP(21);
          if (ct == 0 && t == 1) SS_STAMP(22);
        }
        if (ct == 0) SS_STAMP(23);
      }
    }

    SS_STAMP(1);
    asm volatile("; PHASE_LN1_KV" ::: "memory");
    // ---- norm1 -> LDS; k / v projections of this half's tokens -> exchange buffers ----
    bf16x8_t ring2[3][2];
    {
      SS_PHASE
      ring_fill<2, 3>(ring2, wp + (size_t)(WS_KV + (3 * wave) * (2 * SS_KS)) * 1024, lane);       // lands under the LayerNorm
      layer_norm_to_lds<NW, SS_LN1_VAR>(R, vec + V_N1W, vec + V_N1B, a.eps, smem, wave, lane);
      // the residual is not touched again before proj: its 84 registers go to L2 (a wave-private slab, plain stores) and come back behind the
      // attention -- k / v / q and the attention (13 K fragments resident per head) get the registers
#if SS_PARK
      u32x4_t* const pk = reinterpret_cast<u32x4_t*>(a.park + (size_t)img * PARK_IMG + ((size_t)(half * NW + wave) * 21) * 1024) + lane;
#pragma unroll
      for (int t = 0; t < SS_NT; ++t)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) pk[(t * 3 + ct) * 64] = __builtin_bit_cast(u32x4_t, R[t][ct]);
#endif
    }
    SS_STAMP(2);
    {
      SS_PHASE
#pragma unroll 1
      for (int uu = 0; uu < 3; ++uu) {
        const int u = 3 * wave + uu, h = u >> 1, isv = u & 1;
        const unsigned char* wcur = wp + (size_t)(WS_KV + u * (2 * SS_KS)) * 1024;
        const unsigned char* wnext = uu < 2 ? wcur + (2 * SS_KS) * 1024 : wp + (size_t)(WS_Q + wave * (2 * SS_KS)) * 1024;
        f32x4_t acc[SS_NT][2];
#pragma unroll
        for (int t = 0; t < SS_NT; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
        if (!isv) {
          const float* bk = vec + V_QKVB + SS_C + 32 * h + 4 * g;
          const float4 b0 = *reinterpret_cast<const float4*>(bk), b1 = *reinterpret_cast<const float4*>(bk + 16);
          gemm_unit<2, SS_KS, 3, true>(acc, ring2, wcur, wnext, smem + L_XN, lane);
#pragma unroll
          for (int t = 0; t < SS_NT; ++t) {
            const int kt = half ? (t < 6 ? 7 + t : 13) : t;
            const f32x4_t k0 = {acc[t][0][0] + b0.x, acc[t][0][1] + b0.y, acc[t][0][2] + b0.z, acc[t][0][3] + b0.w};
            const f32x4_t k1 = {acc[t][1][0] + b1.x, acc[t][1][1] + b1.y, acc[t][1][2] + b1.z, acc[t][1][3] + b1.w};
            __builtin_amdgcn_raw_buffer_store_b128(pack_bf8(k0, k1), kr, ((h * 14 + kt) * 64 + lane) * 16, 0, 16);
          }
        } else {
          const float bv0 = vec[V_QKVB + 2 * SS_C + 32 * h + li], bv1 = vec[V_QKVB + 2 * SS_C + 32 * h + 16 + li];
          gemm_unit<2, SS_KS, 3, false>(acc, ring2, wcur, wnext, smem + L_XN, lane);
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              const float bv = dt ? bv1 : bv0;
              const f32x4_t lo = acc[2 * p][dt] + bv;
              u32x4_t pk = {pack_h2(lo[0], lo[1]), pack_h2(lo[2], lo[3]), 0u, 0u};
              if (p < 3) { const f32x4_t hi = acc[2 * p + 1][dt] + bv; pk[2] = pack_h2(hi[0], hi[1]); pk[3] = pack_h2(hi[2], hi[3]); }
              __builtin_amdgcn_raw_buffer_store_b128(pk, vr, ((h * 16 + (half * 4 + p) * 2 + dt) * 64 + lane) * 16, 0, 16);
            }
        }
      }
    }
    SS_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains (R1), then one lane raises the flag
    __syncthreads();
    SS_STAMP(4);
    if (tid == 0) __hip_atomic_store((gu32*)kvflag_mine, (unsigned)(blk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    asm volatile("; PHASE_QPROJ" ::: "memory");
    // ---- q projections of the wave's heads (head `wave`, and head 8 + wave on waves 0..3) ----
    const int nheads = wave0 < NW / 2 ? 2 : 1;
    bf16x8_t Qf[SS_NT];      // (the drain above also waited for the first k-steps of head `wave`: the tail of the last k / v unit fetched them)
    {
      SS_PHASE
#pragma unroll 1
      for (int hu = 0; hu < nheads; ++hu) {
        const int h = wave + NW * hu;
        const unsigned char* wcur = wp + (size_t)(WS_Q + h * (2 * SS_KS)) * 1024;
        const unsigned char* wnext = hu + 1 < nheads ? wp + (size_t)(WS_Q + (h + NW) * (2 * SS_KS)) * 1024 : wcur;
        f32x4_t acc[SS_NT][2];
#pragma unroll
        for (int t = 0; t < SS_NT; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
        const float* bq = vec + V_QKVB + 32 * h + 4 * g;
        const float4 b0 = *reinterpret_cast<const float4*>(bq), b1 = *reinterpret_cast<const float4*>(bq + 16);
        gemm_unit<2, SS_KS, 3, true>(acc, ring2, wcur, wnext, smem + L_XN, lane);
        constexpr float QS = 0.25503486f;      // log2(e) / sqrt(32): softmax(q k^T / sqrt d) as exp2 of the scores (models/lemevit.py:203)
#pragma unroll
        for (int t = 0; t < SS_NT; ++t) {
          const f32x4_t q0 = {(acc[t][0][0] + b0.x) * QS, (acc[t][0][1] + b0.y) * QS, (acc[t][0][2] + b0.z) * QS, (acc[t][0][3] + b0.w) * QS};
          const f32x4_t q1 = {(acc[t][1][0] + b1.x) * QS, (acc[t][1][1] + b1.y) * QS, (acc[t][1][2] + b1.z) * QS, (acc[t][1][3] + b1.w) * QS};
          const u32x4_t qf = pack_bf8(q0, q1);
          if (hu == 0) Qf[t] = as_bf8(qf);
          else *reinterpret_cast<u32x4_t*>(smem + L_H + ((wave * SS_NT + t) * 64 + lane) * 16) = qf;
        }
      }
    }
    SS_STAMP(5);
    if (wave0 == 0) wait_flag(kvflag_peer, (unsigned)(blk + 1), errflag, lane0);
    __syncthreads();          // every q projection has read the LayerNorm output: the attention output may overwrite it; the peer's K / V are published
    SS_STAMP(6);

    asm volatile("; PHASE_ATTN" ::: "memory");
    // ---- attention of the wave's heads -> proj operand in LDS ----
    {
      SS_PHASE
      // 24 (head, query-tile group) units, three per wave: both groups of head `wave`, and one group of head 8 + (wave & 3) -- whose q fragments
      // wave (wave & 3) left in LDS: waves 0..3 take its tiles 0..3, waves 4..7 its tiles 4..6 (11 / 10 query tiles per wave)
      // (round 6, the 8-wave instance: the whole extra head on waves 0 .. 3 -- 14 / 7 query tiles per wave instead of 11 / 10.  The second wave of a SIMD loses the issue arbitration of
      //  this VALU-bound phase: with 11 / 10 waves 0-3 finished in 27 k cycles and idled 12.7 k at the barrier while waves 4-7 took 38 k; now 34 k / 31 k, attention + barrier 40.3 k -> 36.9 k.
      //  The 4-wave instance -- two workgroups per CU -- measured 1 % slower with it and keeps 11 / 10.)
      constexpr bool SS_UNITS_14_7 = NW == 8;
      const int nunits = SS_UNITS_14_7 ? (wave < NW / 2 ? 4 : 2) : 3;
#pragma unroll 1
      for (int u = 0; u < nunits; ++u) {
        const bool third = u >= 2;
        const int h = third ? NW + (wave & (NW / 2 - 1)) : wave;
        const bool groupb = SS_UNITS_14_7 ? (u & 1) : (third ? wave >= NW / 2 : u == 1);
        if (u == 2) {
#pragma unroll
          for (int t = 0; t < SS_NT; ++t) Qf[t] = *reinterpret_cast<const bf16x8_t*>(smem + L_H + (((wave & (NW / 2 - 1)) * SS_NT + t) * 64 + lane) * 16);
        }
        if (!groupb) attn_image<0, 4>(Qf, h, kr, vr, smem, lane);
        else if (half == 0) attn_image<4, 3>(Qf, h, kr, vr, smem, lane);
        else { attn_image<4, 2>(Qf, h, kr, vr, smem, lane); attn_meta(Qf[6], h, kr, vr, smem, lane); }
      }
    }
    SS_STAMP(7);
    bf16x8_t ring3[4][3], ringp[SS_PRD][3];
    {
      SS_PHASE
      ring_fill<3, SS_PRD>(ringp, wp + (size_t)(WS_PROJ + wave * (3 * SS_KS)) * 1024, lane);          // lands under the barrier
#if SS_PARK
      const u32x4_t* const pk = reinterpret_cast<const u32x4_t*>(a.park + (size_t)img * PARK_IMG + ((size_t)(half * NW + wave) * 21) * 1024) + lane;
#pragma unroll
      for (int t = 0; t < SS_NT; ++t)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) R[t][ct] = __builtin_bit_cast(f32x4_t, pk[(t * 3 + ct) * 64]);
#endif
    }
    __syncthreads();

    SS_STAMP(8);
    asm volatile("; PHASE_PROJ" ::: "memory");
    // ---- x += proj(attention) + bias: accumulated ON the residual registers ----
    {
      SS_PHASE
      float4 pbias[3];
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) pbias[ct] = *reinterpret_cast<const float4*>(vec + V_PROJB + 48 * wave + 16 * ct + 4 * g);
      const unsigned char* wcur = wp + (size_t)(WS_PROJ + wave * (3 * SS_KS)) * 1024;
      gemm_unit<3, SS_KS, SS_PRD, true>(R, ringp, wcur, wcur, smem + L_XN, lane);
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int t = 0; t < SS_NT; ++t) { R[t][ct][0] += pbias[ct].x; R[t][ct][1] += pbias[ct].y; R[t][ct][2] += pbias[ct].z; R[t][ct][3] += pbias[ct].w; }
    }

    SS_STAMP(9);
    asm volatile("; PHASE_MLP" ::: "memory");
    // ---- norm2 -> LDS; MLP in 6 chunks of 256 hidden channels; fc2 accumulates on the residual registers ----
    {
      SS_PHASE
      ring_fill<2, 3>(ring2, wp + (size_t)(WS_FC1 + wave * (2 * SS_KS)) * 1024, lane);            // lands under the LayerNorm
      layer_norm_to_lds<NW, SS_LN2_VAR>(R, vec + V_N2W, vec + V_N2B, a.eps, smem, wave, lane);
    }
    SS_STAMP(10);
#pragma unroll 1
    for (int c = 0; c < SS_NCHUNK; ++c) {
      {
        SS_PHASE
        f32x4_t acc[SS_NT][2];
#pragma unroll
        for (int t = 0; t < SS_NT; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
        const unsigned char* wcur = wp + (size_t)(WS_FC1 + (c * NW + wave) * (2 * SS_KS)) * 1024;
        const float* b1p = vec + V_FC1B + 32 * NW * c + 32 * wave + 4 * g;          // (requested ahead of the GEMM: an L2 round trip in the epilogue otherwise)
        const float4 b0 = *reinterpret_cast<const float4*>(b1p), b1 = *reinterpret_cast<const float4*>(b1p + 16);
        // the tail of this unit fetches the first k-steps of the NEXT chunk's fc1 (they ride through this chunk's fc2)
        if (c == 2) SS_STAMP(15);
        gemm_unit<2, SS_KS, 3, true>(acc, ring2, wcur, c + 1 < SS_NCHUNK ? wcur + NW * (2 * SS_KS) * 1024 : wcur, smem + L_XN, lane);
        if (c == 2) SS_STAMP(16);
        ring_fill<3, 4>(ring3, wp + (size_t)(WS_FC2 + (c * NW + wave) * (3 * SS_KSC)) * 1024, lane);      // lands under the GELU pass and the barrier
#pragma unroll
        for (int t = 0; t < SS_NT; ++t) {
          f32x2_t h0 = {acc[t][0][0] + b0.x, acc[t][0][1] + b0.y}, h1 = {acc[t][0][2] + b0.z, acc[t][0][3] + b0.w};
          f32x2_t h2 = {acc[t][1][0] + b1.x, acc[t][1][1] + b1.y}, h3 = {acc[t][1][2] + b1.z, acc[t][1][3] + b1.w};
          gelu4(h0, h1, h2, h3);
          const u32x4_t hf = {pack_bf2(h0[0], h0[1]), pack_bf2(h1[0], h1[1]), pack_bf2(h2[0], h2[1]), pack_bf2(h3[0], h3[1])};
          *reinterpret_cast<u32x4_t*>(smem + L_H + ((wave * SS_NT + t) * 64 + lane) * 16) = hf;
        }
      }
      if (c == 2) SS_STAMP(17);
      __syncthreads();
      if (c == 2) SS_STAMP(18);
      {
        SS_PHASE
        const unsigned char* wcur = wp + (size_t)(WS_FC2 + (c * NW + wave) * (3 * SS_KSC)) * 1024;
        gemm_unit<3, SS_KSC, 4, true>(R, ring3, wcur, wcur, smem + L_H, lane);
      }
      if (c == 2) SS_STAMP(19);
      if (c + 1 < SS_NCHUNK) __syncthreads();
      if (c == 2) SS_STAMP(20);
    }

    SS_STAMP(11);
    asm volatile("; PHASE_BLKEND" ::: "memory");
    // ---- block end: pad slots stay zero; the grid row next to the cut goes to the peer (next block's position embedding) ----
    {
      SS_PHASE
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {        // + mlp.3.bias
        const float4 b = *reinterpret_cast<const float4*>(vec + V_FC2B + 48 * wave + 16 * ct + 4 * g);
#pragma unroll
        for (int t = 0; t < SS_NT; ++t) { R[t][ct][0] += b.x; R[t][ct][1] += b.y; R[t][ct][2] += b.z; R[t][ct][3] += b.w; }
      }
      if (half) {
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) R[5][ct][r] = li >= 4 ? 0.f : R[5][ct][r];
      }
      if (blk + 1 < a.nblocks) {
        const int tok = half ? li : li - 2;
        if ((unsigned)tok < 14u) {
          bf16_t* dst = reinterpret_cast<bf16_t*>(a.halo + (size_t)img * HALO_IMG) + ((size_t)half * 14 + tok) * SS_C + 48 * wave + 4 * g;
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) {
            const f32x4_t v = half ? R[0][ct] : R[6][ct];
            const unsigned long long pk = (unsigned long long)pack_bf2(v[0], v[1]) | ((unsigned long long)pack_bf2(v[2], v[3]) << 32);
            __hip_atomic_store((gu64*)(dst + 16 * ct), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    SS_STAMP(12);
    if (blk + 1 < a.nblocks && tid == 0) __hip_atomic_store((gu32*)haloflag_mine, (unsigned)(blk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- registers -> x_out / c_out ----
  SS_PHASE
#pragma unroll
  for (int t = 0; t < SS_NT; ++t) {
    const int slot = 16 * t + li;
    bf16_t* dst = nullptr;
    if (slot < nimg_slots) dst = a.x_out + ((size_t)img * SS_NIMG + tok0 + slot) * SS_C;
    else if (half && slot >= 96) dst = a.c_out + ((size_t)img * SS_M + slot - 96) * SS_C;
    if (dst) {
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        const float f[4] = {R[t][ct][0], R[t][ct][1], R[t][ct][2], R[t][ct][3]};
        st4(dst + 48 * wave + 16 * ct + 4 * g, f);
      }
    }
  }
}

// ---- packing: reference layouts -> fragment order ------------------------------------------------------------------------------
struct PackArgs { const bf16_t* qkv_w; const bf16_t* proj_w; const bf16_t* fc1_w; const bf16_t* fc2_w; uint4* out; };

// (source matrix, first row, k-step, row stride) of fragment f
template <int NW>
__device__ __forceinline__ void frag_source(int f, const PackArgs& a, const bf16_t*& base, int& row0, int& ks, int& ld) {
  SS_GEO(NW)
  constexpr int UF = 2 * SS_KS, PF = 3 * SS_KS, F2 = 3 * SS_KSC;      // fragments per k / v / q unit, per wave of proj, per (chunk, wave) of fc2
  if (f < WS_Q) { const int u = f / UF, r = f - u * UF; ks = r >> 1; base = a.qkv_w; ld = SS_C; row0 = (1 + (u & 1)) * SS_C + 32 * (u >> 1) + 16 * (r & 1); }
  else if (f < WS_PROJ) { const int q = f - WS_Q, h = q / UF, r = q - h * UF; ks = r >> 1; base = a.qkv_w; ld = SS_C; row0 = 32 * h + 16 * (r & 1); }
  else if (f < WS_FC1) { const int q = f - WS_PROJ, w = q / PF, r = q - w * PF; ks = r / 3; base = a.proj_w; ld = SS_C; row0 = 48 * w + 16 * (r - ks * 3); }
  else if (f < WS_FC2) { const int q = f - WS_FC1, cw = q / UF, r = q - cw * UF; ks = r >> 1; base = a.fc1_w; ld = SS_C; row0 = 32 * NW * (cw / NW) + 32 * (cw % NW) + 16 * (r & 1); }
  else { const int q = f - WS_FC2, cw = q / F2, r = q - cw * F2, ksl = r / 3; ks = SS_KSC * (cw / NW) + ksl; base = a.fc2_w; ld = SS_HID; row0 = 48 * (cw % NW) + 16 * (r - ksl * 3); }
}
template <int NW>
__global__ __launch_bounds__(256) void sstage_pack_kernel(const PackArgs a) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  if (f >= SG<NW>::WS_FRAGS) return;
  const bf16_t* base; int row0, ks, ld;
  frag_source<NW>(f, a, base, row0, ks, ld);
  const bf16_t* src = base + (size_t)(row0 + i) * ld + 32 * ks + 4 * g;
  const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
  a.out[(size_t)f * 64 + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------------------------------
static int ss_waves(int C, int heads, int hidden) {          // NW of the kernel instance that serves this stage, or 0
  for (int nw : {8, 4})
    if (C == 48 * nw && heads == C / 32 && hidden == 4 * C) return nw;
  return 0;
}
template <int NW> static int ss_capacity();
int lmv_sstage_supported(int C, int heads, int hidden, int H, int W, int M, int dtype) {
  const int nw = ss_waves(C, heads, hidden);
  if (!(dtype == LMV_BF16 && nw != 0 && H == SS_G && W == SS_G && M == SS_M)) return 0;
  const int cap = nw == 8 ? ss_capacity<8>() : ss_capacity<4>();
  return cap == 0 || cap >= 9;          // a device that cannot hold 8 incomplete pairs + 1 workgroup (a small partition) takes the per-block schedule
}
size_t lmv_sstage_wpk_bytes(int C, int hidden) { return ss_waves(C, C / 32, hidden) == 4 ? (size_t)SG<4>::WS_FRAGS * 1024 : (size_t)SG<8>::WS_FRAGS * 1024; }
size_t lmv_sstage_vec_floats(int C, int hidden) { (void)hidden; return (size_t)23 * C; }
template <int NW> static size_t ss_flag_bytes(int B) { return ((size_t)(4 * B + 8 + 1) * 4 + 1023) & ~(size_t)1023; }          // kv | halo flags, the 8 ticket counters
template <int NW> static size_t ss_workspace(int B) {
  return ss_flag_bytes<NW>(B) + (size_t)B * (SG<NW>::KBUF_IMG + SG<NW>::VBUF_IMG + SG<NW>::HALO_IMG + SG<NW>::PARK_IMG);
}
size_t lmv_sstage_workspace_bytes(int B, int C) { return C == 192 ? ss_workspace<4>(B) : ss_workspace<8>(B); }
// Workgroups of the instance the device holds at once (occupancy query x CUs, cached per device; 0: no device / unknown).  Both halves of an image wait for each other, so a launch
// takes capacity / 2 images (whole groups of 8; a larger batch runs as consecutive launches).  The slot assignment is by ticket (stage_ticket), so these counts are a matter of speed:
// what progress needs is room for 8 incomplete pairs + 1 workgroup per launch in flight -- lmv_sstage_supported / lmv_sstage_max_concurrent.
template <int NW> static int ss_capacity() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int v = cache[dev & 63].load(std::memory_order_relaxed);
  if (!v) { v = lmv_stage_capacity(reinterpret_cast<const void*>(sstage_kernel<NW>), 64 * NW, SG<NW>::L_TOTAL); cache[dev & 63].store(v, std::memory_order_relaxed); }
  return v;
}
template <int NW> static int ss_max_images() {
  const int cap = ss_capacity<NW>();
  if (cap <= 0) return NW == 8 ? 128 : 256;          // (no device in this process: the MI355X figure, for workspace sizing only)
  const int n = cap / 2 / 8 * 8;
  return n < 8 ? 8 : n;
}
int lmv_sstage_max_images(int C) { return C == 192 ? ss_max_images<4>() : ss_max_images<8>(); }
int lmv_sstage_max_concurrent(int C) {
  int cap = C == 192 ? ss_capacity<4>() : ss_capacity<8>();
  if (cap <= 0) cap = C == 192 ? 512 : 256;          // (no device in this process: the MI355X figure)
  return (cap - 1) / 8;
}

int lmv_sstage_pack(const lmv_sstage_block_params* p, void* wpk_out, float* vec_out, void* stream) {
  if (!p || !wpk_out || !vec_out) LMV_FAIL(LMV_ERR_SHAPE, "sstage_pack: null argument");
  const int nw = ss_waves(p->C, p->heads, p->hidden);
  if (!nw) LMV_FAIL(LMV_ERR_DTYPE, "sstage_pack: C = %d / heads = %d / hidden = %d is not a supported stage", p->C, p->heads, p->hidden);
  const void* ptrs[] = {p->qkv_w, p->proj_w, p->fc1_w, p->fc2_w, p->n1_w, p->n1_b, p->qkv_b, p->proj_b, p->n2_w, p->n2_b, p->fc1_b, p->fc2_b, p->pos_w, p->pos_b};
  for (const void* q : ptrs) if (!q || !lmv_aligned16(q)) LMV_FAIL(LMV_ERR_SHAPE, "sstage_pack: null or misaligned parameter pointer");
  if (!lmv_aligned16(wpk_out) || !lmv_aligned16(vec_out)) LMV_FAIL(LMV_ERR_SHAPE, "sstage_pack: misaligned output");
  hipStream_t st = (hipStream_t)stream;
  PackArgs a{(const bf16_t*)p->qkv_w, (const bf16_t*)p->proj_w, (const bf16_t*)p->fc1_w, (const bf16_t*)p->fc2_w, (uint4*)wpk_out};
  if (nw == 8) hipLaunchKernelGGL(sstage_pack_kernel<8>, dim3((SG<8>::WS_FRAGS + 3) / 4), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(sstage_pack_kernel<4>, dim3((SG<4>::WS_FRAGS + 3) / 4), dim3(256), 0, st, a);
  LMV_CHECK_LAUNCH("sstage_pack");
  const int C = p->C;
  const struct { const float* src; int off, n; } v[] = {{p->n1_w, 0, C}, {p->n1_b, C, C}, {p->qkv_b, 2 * C, 3 * C}, {p->proj_b, 5 * C, C}, {p->n2_w, 6 * C, C},
                                                        {p->n2_b, 7 * C, C}, {p->fc1_b, 8 * C, 4 * C}, {p->fc2_b, 12 * C, C}, {p->pos_b, 22 * C, C}};
  for (const auto& e : v)
    if (hipMemcpyAsync(vec_out + e.off, e.src, (size_t)e.n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "sstage_pack: vector copy failed");
  hipLaunchKernelGGL(stage_posw_pack_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, st, p->pos_w, reinterpret_cast<unsigned*>(vec_out + 13 * C), 9 * C);          // the 3 x 3 taps as bf16 pair words
  LMV_CHECK_LAUNCH("sstage_pack");
  return LMV_OK;
}

template <int NW>
static int ss_launch(const lmv_sstage_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* workspace, size_t workspace_bytes, hipStream_t st) {
  using G = SG<NW>;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sstage_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, G::L_TOTAL) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "sstage_fwd: cannot reserve LDS");
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  // Both halves of an image wait for each other: a launch takes as many images as the device holds pairs of workgroups (MI355X: 128 at 8 waves, 256 at 4); a larger batch runs as
  // consecutive launches over ranges of images.
  const int MAXB = ss_max_images<NW>();
  unsigned* const errword = lmv_stage_errword(st);
  if (!errword) LMV_FAIL(LMV_ERR_LAUNCH, "sstage_fwd: cannot allocate the error word (the first stage call of a process must not be inside a stream capture)");
  for (int b0 = 0; b0 < d->B; b0 += MAXB) {
    const int nb = d->B - b0 < MAXB ? d->B - b0 : MAXB;
    if (workspace_bytes < ss_workspace<NW>(nb)) LMV_FAIL(LMV_ERR_WORKSPACE, "sstage_fwd: workspace too small");
    const size_t flags = ss_flag_bytes<NW>(nb);
    unsigned char* ws = (unsigned char*)workspace;
    if (hipMemsetAsync(ws, 0, flags, st) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "sstage_fwd: flag reset failed");      // every polled word, every call (Guideline 16)
    SsArgs a{};
    a.x_in = (const bf16_t*)x + (size_t)b0 * SS_NIMG * G::C; a.c_in = (const bf16_t*)c + (size_t)b0 * SS_M * G::C;
    a.x_out = (bf16_t*)x_out + (size_t)b0 * SS_NIMG * G::C; a.c_out = (bf16_t*)c_out + (size_t)b0 * SS_M * G::C;
    a.wpk = (const uint4*)d->wpk; a.vec = d->vec;
    a.flags = (unsigned*)ws; a.kbuf = ws + flags; a.vbuf = a.kbuf + (size_t)nb * G::KBUF_IMG; a.halo = a.vbuf + (size_t)nb * G::VBUF_IMG; a.park = a.halo + (size_t)nb * G::HALO_IMG;
    a.B = nb; a.nblocks = d->nblocks; a.eps = d->eps;
    a.timing = (unsigned long long*)d->timing; a.timing_block = d->timing_block;
    a.err = errword;
    a.tickets = a.flags + 4 * nb; a.quota = 2u * (unsigned)((nb + 7) / 8); a.skew = (unsigned)lmv_config().stage_ticket_skew;
    const int nwg = 2 * ((nb + 7) / 8) * 8;          // = 8 quota
    hipLaunchKernelGGL(sstage_kernel<NW>, dim3(nwg), dim3(64 * NW), G::L_TOTAL, st, a);
    LMV_CHECK_LAUNCH("sstage_fwd");
  }
  return LMV_OK;
}

int lmv_sstage_fwd(const lmv_sstage_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !x || !c || !x_out || !c_out || !workspace) LMV_FAIL(LMV_ERR_SHAPE, "sstage_fwd: null argument");
  if (!lmv_sstage_supported(d->C, d->heads, d->hidden, d->H, d->W, d->M, d->dtype)) LMV_FAIL(LMV_ERR_DTYPE, "sstage_fwd: unsupported stage shape / dtype");
  if (d->B <= 0 || d->nblocks <= 0 || !d->wpk || !d->vec) LMV_FAIL(LMV_ERR_SHAPE, "sstage_fwd: bad descriptor");
  const void* ptrs[] = {x, c, x_out, c_out, workspace, d->wpk, d->vec};
  for (const void* q : ptrs) if (!lmv_aligned16(q)) LMV_FAIL(LMV_ERR_SHAPE, "sstage_fwd: pointers must be 16-byte aligned");
  // (timing probe: the Linear layers of 212 tokens per image + the two attention products of every head; bytes: tokens in and out once, the weights once)
  const double Cd = d->C, tok = (double)d->H * d->W + d->M, img = (double)d->H * d->W;
  LmvTimedLaunch timed(stream, (double)d->B * d->nblocks * (tok * Cd * 12.0 * Cd * 2.0 + (Cd / 32.0) * (img * img + (double)d->M * d->M) * 32.0 * 4.0),
                       (double)d->B * tok * Cd * 2.0 * 2.0 + (double)d->nblocks * 12.0 * Cd * Cd * 2.0, 1);
  return ss_waves(d->C, d->heads, d->hidden) == 8 ? ss_launch<8>(d, x, c, x_out, c_out, workspace, workspace_bytes, (hipStream_t)stream)
                                                 : ss_launch<4>(d, x, c, x_out, c_out, workspace, workspace_bytes, (hipStream_t)stream);
}
