// dstage.hip -- a whole run of "D" blocks (Dual Cross-Attention: stages 1 / 2 of LeMeViT) as ONE persistent launch with the token rows resident on
// chip (round 4; inference form).  The construction of csrc/sstage.hip carried to the blocks whose image tokens do not fit one workgroup.
//
// Reference math: LeMeBlock.forward_with_xc, models/lemevit.py:542-582 (live branch :560-564): x = x + dwconv3x3(x) (:546);
// (ax, ac) = DualCrossAttention(norm1(x), norm1(c)); x = x + ax; x = x + mlp(norm2(x)); c = c + ac; c = c + mlp(norm2(c)), with
// DualCrossAttention :220-324 (live :288-302): qkv1 = W1 x + b1, qkv2 = W2 c + b2, ax = proj_x(softmax(q1 k2^T s_x) v2) (every image token against the
// 16 meta keys), ac = proj_c(softmax(q2 k1^T s_c) v1) (16 meta queries against ALL image keys), s_x = log_N(M) C^-1/2, s_c = C^-1/2 (:235,255-256).
//
// Decomposition.  An image's N = GW x GW image tokens are KWG workgroups of 112 tokens (7 tiles of 16 = ROWS whole grid rows: 4 rows of 28 at stage 2,
// 2 rows of 56 at stage 1) plus ONE meta workgroup that owns the 16 meta tokens.  All KWG + 1 workgroups of an image are resident at the same time and walk
// the blocks of the stage together; a launch processes `slots` images at a time and loops over the batch.  What crosses workgroups, per block, through L2
// (write-through stores, sc1 loads, one relaxed agent-scope flag per hand-off, epochs counted inside the call -- cdna_hip_programming.md Guideline 16, R1):
//   * the grid rows next to a cut, both ways (3 x 3 position embedding), double-buffered by block parity;
//   * meta -> image: the operand fragments the image tokens need from the meta tokens -- K2 and V2 of every head (x-direction attention, 16 keys) and
//     q~ = (q2 W_k1[h]) s_c log2 e: the meta queries pushed through the k-projection of the image tokens, so that the c-direction scores are q~ . norm1(x)
//     straight off the LayerNorm output in LDS -- the image tokens' k projection (C x C per token) is never computed; its bias shifts every score of a
//     query by the same amount and drops out of the softmax;
//   * image -> meta: per head the partial (max, sum, sum p v) of the 16 meta queries over the workgroup's 112 keys; the meta workgroup combines the KWG
//     partials (log-sum-exp), adds the v bias (sum p = 1), and runs the whole c path (proj_c, norm2, MLP on 16 tokens) itself.
// Inside an image workgroup: exactly the data flow of sstage.hip (NW waves x 48 channels: C = 48 NW; residual in registers; weights straight from L2 in fragment
// order; token operands in LDS in fragment order; transposed outputs as ready operand fragments; MLP in 6 chunks); v1 is computed with the operands swapped
// (the V^T operand of P V) right behind the scores of its head, in the same wave.
#include <atomic>
#include "stage_common.h"

#ifndef DS_STG_ENTRY
#define DS_STG_ENTRY 40      // bytes per entry of the dwconv staging image (32 of channels; DG::STG_E)
#endif
namespace {

constexpr int DS_M = 16, DS_NSTAMP = 16;

// NW waves per workgroup, a wave owns CT channel tiles of 16 (CT = 3: C = 96 / 192, LeMeViT-Base / -Small; CT = 2: C = 64 / 128, LeMeViT-Tiny); GW x GW image tokens
template <int NW, int GW, int CTILES> struct DG {
  static constexpr int CT = CTILES, CW = 16 * CT, C = CW * NW, NH = C / 32, HID = 4 * C, KS = C / 32, KSC = NW, NCHUNK = 2 * CT;
  static constexpr int PRD = KS % 4 == 0 ? 4 : 3;                                  // ring depth of the proj units (CT tiles per k-step; KS = 2: the whole unit up front)
  static constexpr int URD = KS % 3 == 0 ? 3 : (KS % 4 == 0 ? 4 : 3);              // ring depth of the 2-tile units (q1, v1, k2 / v2 / q2, fc1)
  static_assert((CT * NW) % 2 == 0, "whole heads");
  static constexpr int NT = GW % 7 == 0 ? 7 : 6, TOK = 16 * NT;                   // token tiles / tokens of an image workgroup: 112 (56 x 56, 28 x 28 grids) or 96 (96 x 96, 48 x 48: 384 x 384 images)
  static constexpr int ROWS = TOK / GW, KWG = GW / ROWS, NIMG = GW * GW, NWG = KWG + 1;
  static constexpr int CSUB = KWG % 2 == 0 ? 2 : 1, NWG_C = KWG / CSUB + 1;      // "C" blocks: row groups per image workgroup, workgroups per slot
  static_assert(ROWS * GW == TOK && KWG * ROWS == GW, "an image workgroup is whole grid rows");
  // packed weights of a block, 1 KB fragments (stage_common.h / sstage.hip: lane (g, i) holds W[row0 + i][32 ks + 16 (j >> 2) + 4 g + (j & 3)])
  static constexpr int WS_Q1 = 0;                                   // [head][ks][n 2]: rows 32 h + 16 n of qkv1.weight
  static constexpr int WS_V1 = WS_Q1 + NH * KS * 2;                 // [head][ks][n 2]: rows 2 C + 32 h + 16 n of qkv1.weight
  static constexpr int WS_PX = WS_V1 + NH * KS * 2;                 // [wave][ks][n CT]: rows CW w + 16 n of proj_x.weight
  static constexpr int WS_FC1 = WS_PX + NW * KS * CT;                // [chunk][wave][ks][n 2]
  static constexpr int WS_FC2 = WS_FC1 + NCHUNK * NW * KS * 2;      // [chunk][wave][ksl KSC][n CT]
  static constexpr int WS_K2 = WS_FC2 + NCHUNK * NW * KSC * CT;     // meta: [head][ks][n 2]: rows C + 32 h + 16 n of qkv2.weight
  static constexpr int WS_V2 = WS_K2 + NH * KS * 2;                 //       rows 2 C + 32 h + 16 n
  static constexpr int WS_Q2 = WS_V2 + NH * KS * 2;                 //       rows 32 h + 16 n
  static constexpr int WS_K1T = WS_Q2 + NH * KS * 2;                // [head][m < C / 16]: lane (g, i) holds qkv1.weight[C + 32 h + 16 (j >> 2) + 4 g + (j & 3)][16 m + i]
  static constexpr int WS_PC = WS_K1T + NH * (C / 16);              // [wave][ks][n CT]: proj_c.weight
  static constexpr int WS_K1 = WS_PC + NW * KS * CT;                // [head][ks][n 2]: rows C + 32 h + 16 n of qkv1.weight ("S" blocks: the image tokens' keys are computed)
  static constexpr int WS_FRAGS = WS_K1 + NH * KS * 2;
  static constexpr int V_N1W = 0, V_N1B = C, V_QKV1B = 2 * C, V_QKV2B = 5 * C, V_PXB = 8 * C, V_PCB = 9 * C, V_N2W = 10 * C, V_N2B = 11 * C, V_FC1B = 12 * C,
                       V_FC2B = 16 * C, V_POSW = 17 * C, V_POSB = 26 * C, V_FLOATS = 27 * C;
  // LDS of an image workgroup (the meta workgroup uses the front of the same regions with one token tile)
  // L_H | L_XN in this order: the attention output (the proj_x operand, KS k-steps) is written from L_H on -- the heads of the first NW k-steps land in L_H while the
  // other waves still read the LayerNorm output in L_XN, only the last KS - NW wait (in registers) for the barrier
  static constexpr int L_H = 0, L_H_BYTES = KSC * NT * 1024;
  static constexpr int L_XN = L_H_BYTES, L_XN_BYTES = KS * NT * 1024;
  static constexpr int L_AO = L_H;
  static constexpr int L_STAT = L_XN + L_XN_BYTES, L_STAT_BYTES = NW * TOK * 8;
  static constexpr int L_TOTAL = L_STAT + L_STAT_BYTES;
  // dwconv staging: [ROWS + 2][GW + 2] entries of 16 channels, bf16: 32 bytes in a 40-byte slot (round 5; csrc/sstage.hip, SS_STG_ROW: with packed entries the 16 lanes of a
  // ds_read2_b64 lane group hit 4 banks, 8 p mod 32; 10 p mod 32 runs through all 16 even banks).  Every access to the image is 8 bytes wide.
  static constexpr int STG_E = DS_STG_ENTRY;
  static constexpr int STG_COLS = GW + 2, STG_ENT = (ROWS + 2) * STG_COLS, STG_WAVE = (STG_ENT * STG_E + 15) / 16 * 16;
  static_assert(NW * STG_WAVE <= L_STAT, "staging overlaps the statistics");
  // per-slot workspace
  static constexpr int MF_HEAD = 3 + KS;                                   // meta fragments per head: K2 | V2 d-tile 0 | V2 d-tile 1 | q~ [ks]
  static constexpr size_t MFRAG_BYTES = (size_t)2 * NH * MF_HEAD * 1024;   // [parity][head][MF_HEAD]
  static constexpr size_t HALO_BYTES = (size_t)2 * KWG * 2 * GW * C * 2;   // [parity][workgroup][first | last grid row][GW tokens][C] bf16
  static constexpr size_t PART_BYTES = (size_t)KWG * NH * 3 * 1024;        // [workgroup][head][O d-tile 0 | O d-tile 1 | (max, sum)]: f32x4 per lane
  // "S" blocks (KIND 2, the 8-wave instance only): K and V^T operand fragments of all image tokens of the image, [parity][K | V][head][KWG NT key tiles] x 1 KB
  static constexpr size_t KV_HALF = (size_t)NH * KWG * NT * 1024, KV_BYTES = GW == 24 ? 4 * KV_HALF : 0;
  static constexpr size_t SLOT_BYTES = MFRAG_BYTES + HALO_BYTES + PART_BYTES + KV_BYTES;
  static constexpr int FLAGS_PER_SLOT = 3 * KWG + 1;                       // halo [KWG] | partial [KWG] | meta | K / V ready [KWG] ("S" blocks)
};

struct DsArgs {
  const bf16_t* x_in; const bf16_t* c_in; bf16_t* x_out; bf16_t* c_out;
  const uint4* wpk; const float* vec;
  unsigned char* slots; unsigned* flags; unsigned* err;          // flags: [nslots][FLAGS_PER_SLOT] | [8] tickets; err: the sticky error word (lmv_stage_error_count)
  unsigned* tickets; unsigned quota, skew;                            // stage_ticket (stage_common.h): 8 counters of `quota` = nslots / 8 x (workgroups per slot) tickets
  int B, nblocks, nslots; float eps, sx, sc;       // sx / sc: the two attention scales times log2 e
  unsigned long long* timing; int timing_block;
};

#define DS_STAMP(k)                                                                                                                              \
  do {                                                                                                                                           \
    if (a.timing && gb == a.timing_block && lane == 0) a.timing[((size_t)blockIdx.x * NW + wave) * DS_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

// LayerNorm of NTT register-resident token tiles -> bf16 operand fragments at `xn` ([k-step][tile]); statistics over the NW waves through `stat`
template <int NW, int NTT, int CT>
__device__ __forceinline__ void ds_layer_norm(const f32x4_t (&R)[NTT][CT], const float* gam, const float* bet, float eps, unsigned char* xn, float2* stat, int wave, int lane) {
  __builtin_amdgcn_sched_barrier(0); asm volatile("; LN_FENCE_BEGIN" ::: "memory");          // (tests/test_isa_cpu.py: no packed fp32 arithmetic between the markers: the statistics and their combine)
  constexpr int CW = 16 * CT, C = CW * NW;
  const int g = lane >> 4, li = lane & 15;
  // statistics without cancellation: per wave the mean and the centred sum of squares of its CW channels (two passes over registers), combined over the waves with the
  // parallel-variance formula -- see layer_norm_to_lds (csrc/sstage.hip)
#pragma unroll
  for (int t = 0; t < NTT; ++t) {
    float s1 = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) s1 += R[t][ct][r];
    const float mw = xsum4(s1) * (1.f / CW);
    float m2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float dlt = R[t][ct][r] - mw; m2 = fmaf(dlt, dlt, m2); }
    m2 = xsum4(m2);
    if (g == 0) stat[wave * (16 * NTT) + t * 16 + li] = make_float2(mw, m2);
  }
  float4 ga[CT], be[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { ga[ct] = *reinterpret_cast<const float4*>(gam + CW * wave + 16 * ct + 4 * g); be[ct] = *reinterpret_cast<const float4*>(bet + CW * wave + 16 * ct + 4 * g); }
  __syncthreads();
  float mean[NTT], rstd[NTT];
#pragma unroll
  for (int t = 0; t < NTT; ++t) {
    float2 p[NW];
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { p[w] = stat[w * (16 * NTT) + t * 16 + li]; sm += p[w].x; sq += p[w].y; }
    asm volatile("" : "+v"(sm), "+v"(sq));          // scalar code on purpose: the SLP-packed form of this combine is not run-to-run reproducible (csrc/sstage.hip, SS_LN_PACKED)
    mean[t] = sm * (1.f / NW);
    float dev = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const float dlt = p[w].x - mean[t]; dev = fmaf(dlt, dlt, dev); }
    rstd[t] = rsqrtf(fmaf(dev, (float)CW, sq) * (1.f / C) + eps);
    asm volatile("" : "+v"(mean[t]), "+v"(rstd[t]));
  }
asm volatile("; LN_FENCE_END" ::: "memory"); __builtin_amdgcn_sched_barrier(0);          // (the normalisation below packs two CHANNELS of a token with VGPR-pair splats: that form has always been in the bit-stable builds)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int T = CT * wave + ct;
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
      const float y0 = fmaf((R[t][ct][0] - mean[t]) * rstd[t], ga[ct].x, be[ct].x), y1 = fmaf((R[t][ct][1] - mean[t]) * rstd[t], ga[ct].y, be[ct].y);
      const float y2 = fmaf((R[t][ct][2] - mean[t]) * rstd[t], ga[ct].z, be[ct].z), y3 = fmaf((R[t][ct][3] - mean[t]) * rstd[t], ga[ct].w, be[ct].w);
      *reinterpret_cast<uint2*>(xn + (((T >> 1) * NTT + t) * 64 + lane) * 16 + (T & 1) * 8) = make_uint2(pack_bf2(y0, y1), pack_bf2(y2, y3));
    }
  }
  __syncthreads();
}

// the MLP half on NTT register-resident token tiles: fc1 chunk -> GELU -> LDS -> fc2 partial sums on R (bias of fc2 added by the caller)
template <int NW, int NTT, typename G>
__device__ __forceinline__ void ds_mlp(f32x4_t (&R)[NTT][G::CT], const unsigned char* wp, const float* vec, unsigned char* xn, unsigned char* hb, int lane0, int wave0) {
  constexpr int KS = G::KS, KSC = G::KSC, CT = G::CT, URD = G::URD, RD2 = KSC % 4 == 0 ? 4 : 3;
  constexpr bool PRE1 = KS % URD == 0;          // the fc1 fragments of chunk c + 1 are fetched behind those of chunk c (else: a unit is loaded whole, up front)
  bf16x8_t ring2[URD][2], ring3[RD2][CT];
  if constexpr (PRE1) ring_fill<2, URD>(ring2, wp + (size_t)(G::WS_FC1 + wave0 * (2 * KS)) * 1024, lane0);
#pragma unroll 1
  for (int c = 0; c < G::NCHUNK; ++c) {
    {
      __builtin_amdgcn_sched_barrier(0);
      int lane = lane0; asm volatile("" : "+v"(lane));
      int wave = wave0; asm volatile("" : "+s"(wave));
      const int g = lane >> 4;
      f32x4_t acc[NTT][2];
#pragma unroll
      for (int t = 0; t < NTT; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
      const unsigned char* wcur = wp + (size_t)(G::WS_FC1 + (c * NW + wave) * (2 * KS)) * 1024;
      const float* b1p = vec + G::V_FC1B + 32 * NW * c + 32 * wave + 4 * g;
      const float4 b0 = *reinterpret_cast<const float4*>(b1p), b1 = *reinterpret_cast<const float4*>(b1p + 16);
      if constexpr (!PRE1) ring_fill<2, URD>(ring2, wcur, lane);
      gemm_unit<2, KS, URD, true, NTT>(acc, ring2, wcur, c + 1 < G::NCHUNK ? wcur + NW * (2 * KS) * 1024 : wcur, xn, lane);
      ring_fill<CT, RD2>(ring3, wp + (size_t)(G::WS_FC2 + (c * NW + wave) * (CT * KSC)) * 1024, lane);
#pragma unroll
      for (int t = 0; t < NTT; ++t) {
        f32x2_t h0 = {acc[t][0][0] + b0.x, acc[t][0][1] + b0.y}, h1 = {acc[t][0][2] + b0.z, acc[t][0][3] + b0.w};
        f32x2_t h2 = {acc[t][1][0] + b1.x, acc[t][1][1] + b1.y}, h3 = {acc[t][1][2] + b1.z, acc[t][1][3] + b1.w};
        gelu4(h0, h1, h2, h3);
        const u32x4_t hf = {pack_bf2(h0[0], h0[1]), pack_bf2(h1[0], h1[1]), pack_bf2(h2[0], h2[1]), pack_bf2(h3[0], h3[1])};
        *reinterpret_cast<u32x4_t*>(hb + ((wave * NTT + t) * 64 + lane) * 16) = hf;
      }
    }
    __syncthreads();
    {
      __builtin_amdgcn_sched_barrier(0);
      int lane = lane0; asm volatile("" : "+v"(lane));
      int wave = wave0; asm volatile("" : "+s"(wave));
      const unsigned char* wcur = wp + (size_t)(G::WS_FC2 + (c * NW + wave) * (CT * KSC)) * 1024;
      gemm_unit<CT, KSC, RD2, true, NTT>(R, ring3, wcur, wcur, hb, lane);
    }
    if (c + 1 < G::NCHUNK) __syncthreads();
  }
}

template <int NW, int GW, int CT, int KIND>          // KIND: 0 "D" blocks, 1 "C" blocks, 2 "S" blocks (self-attention of the image tokens across the workgroups of an image; NW = 8)
__global__ __launch_bounds__(64 * NW, 2) void dstage_kernel(const DsArgs a) {
  using G = DG<NW, GW, CT>;
  constexpr int C = G::C, NH = G::NH, KS = G::KS, KWG = G::KWG, ROWS = G::ROWS, CW = G::CW, URD = G::URD, NT = G::NT, TOK = G::TOK, NP = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63, wave0 = __builtin_amdgcn_readfirstlane(tid >> 6);
#define DS_PHASE                                                    \
  __builtin_amdgcn_sched_barrier(0);                                \
  int lane = lane0; asm volatile("" : "+v"(lane));                  \
  int wave = wave0; asm volatile("" : "+s"(wave));                  \
  const int g = lane >> 4, li = lane & 15; (void)g; (void)li; (void)wave;
  // (slot, role) by ticket, not by blockIdx: the workgroups that are resident always form complete slots + at most one incomplete slot per counter, under any dispatch order
  // (stage_ticket, stage_common.h); the workgroups of a slot take consecutive tickets of the counter of the XCD they run on and share its L2 (a speed matter only)
  const int tk = stage_ticket(a.tickets, a.quota, smem, a.skew);
  if (tk < 0) return;
  const int xcd = tk & 7, jj = tk >> 3;
  // "D" blocks: KWG image workgroups + the meta workgroup per slot; "C" blocks (the image side of a block is a third of the work and every block starts from x_in): an image
  // workgroup takes SUB = 2 row groups in turn, KWG / 2 + 1 workgroups per slot, twice the slots in flight
  constexpr int sub_n = KIND == 1 ? G::CSUB : 1, nimgwg = KWG / sub_n, nwg = nimgwg + 1;
  const int slot = (jj / nwg) * 8 + xcd, rolep = jj % nwg;               // rolep < nimgwg: image rows; rolep == nimgwg: the meta tokens
  if (slot >= a.nslots) return;
  unsigned* const fl = a.flags + (size_t)slot * G::FLAGS_PER_SLOT;
  unsigned* const haloflag = fl, * const partflag = fl + KWG, * const mflag = fl + 2 * KWG, * const kvflag = fl + 2 * KWG + 1;
  unsigned* const errflag = a.err;
  unsigned char* const sb = a.slots + (size_t)slot * G::SLOT_BYTES;
  unsigned char* const mfr = sb, * const halo = sb + G::MFRAG_BYTES, * const part = halo + G::HALO_BYTES, * const kvb = part + G::PART_BYTES;
  (void)kvb; (void)kvflag;
  float2* const stat = reinterpret_cast<float2*>(smem + G::L_STAT);

  int round = 0;
#pragma unroll 1
  for (int img = slot; img < a.B; img += a.nslots, ++round) {
    if (rolep == nimgwg) {
      asm volatile("; PHASE_META");
      // =================================== the meta workgroup: the 16 meta tokens of image `img` ===================================
      f32x4_t Rc[1][CT];
      {
        DS_PHASE
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { float f[4]; ld4(a.c_in + ((size_t)img * DS_M + li) * C + CW * wave + 16 * ct + 4 * g, f); Rc[0][ct] = f32x4_t{f[0], f[1], f[2], f[3]}; }
      }
#pragma unroll 1
      for (int blk = 0; blk < a.nblocks; ++blk) {
        const int gb = round * a.nblocks + blk;                    // global block counter of this slot: epochs and buffer parities
        const int lane = lane0, wave = wave0;                      // (stamps only)
        DS_STAMP(8);
        const unsigned char* const wp = reinterpret_cast<const unsigned char*>(a.wpk) + (size_t)blk * G::WS_FRAGS * 1024;
        const float* const vec = a.vec + (size_t)blk * G::V_FLOATS;
        unsigned char* const mf = mfr + (size_t)(gb & 1) * (G::MFRAG_BYTES / 2);
        const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(mf, 0, (int)(G::MFRAG_BYTES / 2), 0x00020000);
        // ---- norm1(c) -> LDS; K2 / V2 / q2 of every head; q~ = q2 W_k1; all published for the image workgroups ----
        {
          DS_PHASE
          ds_layer_norm<NW, 1, CT>(Rc, vec + G::V_N1W, vec + G::V_N1B, a.eps, smem + G::L_XN, stat, wave, lane);
        }
        DS_STAMP(9);
        {
          DS_PHASE
          // 3 NH units (k2, v2, q2 of a head: 2 channel tiles x 1 token tile x KS) dealt round-robin to the waves; q2 fragments also go to LDS for the q~ pass
          bf16x8_t ring[URD][2];
#pragma unroll 1
          for (int u = wave; u < 3 * NH; u += NW) {
            const int h = u / 3, typ = u - 3 * h;                 // 0: k2, 1: v2, 2: q2
            if (KIND == 1 && typ != 2) continue;                // "C" blocks: no x-direction
            const unsigned char* wcur = wp + (size_t)((typ == 0 ? G::WS_K2 : typ == 1 ? G::WS_V2 : G::WS_Q2) + h * (2 * KS)) * 1024;
            ring_fill<2, URD>(ring, wcur, lane);
            f32x4_t acc[1][2] = {{f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}}};
            if (typ == 1) {
              const float bv0 = vec[G::V_QKV2B + 2 * C + 32 * h + li], bv1 = vec[G::V_QKV2B + 2 * C + 32 * h + 16 + li];
              gemm_unit<2, KS, URD, false, 1>(acc, ring, wcur, wcur, smem + G::L_XN, lane);
#pragma unroll
              for (int dt = 0; dt < 2; ++dt) {
                const f32x4_t lo = acc[0][dt] + (dt ? bv1 : bv0);
                const u32x4_t pk = {pack_h2(lo[0], lo[1]), pack_h2(lo[2], lo[3]), 0u, 0u};
                if constexpr (KIND == 2) *reinterpret_cast<u32x4_t*>(smem + G::L_XN + ((KS + NH + 2 * h + dt) * 64 + lane) * 16) = pk;          // ("S" blocks: the meta tokens attend among themselves, here)
                else __builtin_amdgcn_raw_buffer_store_b128(pk, mr, ((h * G::MF_HEAD + 1 + dt) * 64 + lane) * 16, 0, 16);
              }
            } else {
              const float* bp = vec + G::V_QKV2B + (typ == 0 ? C : 0) + 32 * h + 4 * g;
              const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 16);
              gemm_unit<2, KS, URD, true, 1>(acc, ring, wcur, wcur, smem + G::L_XN, lane);
              const float sc = typ == 0 ? 1.f : a.sc;
              const f32x4_t k0 = {(acc[0][0][0] + b0.x) * sc, (acc[0][0][1] + b0.y) * sc, (acc[0][0][2] + b0.z) * sc, (acc[0][0][3] + b0.w) * sc};
              const f32x4_t k1 = {(acc[0][1][0] + b1.x) * sc, (acc[0][1][1] + b1.y) * sc, (acc[0][1][2] + b1.z) * sc, (acc[0][1][3] + b1.w) * sc};
              const u32x4_t pk = pack_bf8(k0, k1);
              if (typ == 0) {
                if constexpr (KIND == 2) *reinterpret_cast<u32x4_t*>(smem + G::L_XN + ((KS + h) * 64 + lane) * 16) = pk;
                else __builtin_amdgcn_raw_buffer_store_b128(pk, mr, ((h * G::MF_HEAD) * 64 + lane) * 16, 0, 16);
              } else *reinterpret_cast<u32x4_t*>(smem + G::L_H + (h * 64 + lane) * 16) = pk;
            }
          }
        }
        __syncthreads();
        DS_STAMP(10);
        if constexpr (KIND == 2) {
          // "S" blocks: softmax(q2 k2^T / sqrt 32) v2 per head on the 16 meta tokens -> the proj_c operand (the LayerNorm output in L_XN is dead behind the barrier above)
          DS_PHASE
#pragma unroll 1
          for (int h = wave; h < NH; h += NW) {
            const bf16x8_t q2f = *reinterpret_cast<const bf16x8_t*>(smem + G::L_H + (h * 64 + lane) * 16);
            const bf16x8_t k2f = *reinterpret_cast<const bf16x8_t*>(smem + G::L_XN + ((KS + h) * 64 + lane) * 16);
            const f16x8_t v20 = *reinterpret_cast<const f16x8_t*>(smem + G::L_XN + ((KS + NH + 2 * h) * 64 + lane) * 16);
            const f16x8_t v21 = *reinterpret_cast<const f16x8_t*>(smem + G::L_XN + ((KS + NH + 2 * h + 1) * 64 + lane) * 16);
            const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4_t sc = mfma_bf16(k2f, q2f, z4);          // S^T[key][query]
            const float m = xmax4(max4(sc));
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(sc[r] - m);
            const float inv = 1.f / xsum4((e[0] + e[1]) + (e[2] + e[3]));
            const u32x4_t pk = {pack_h2(e[0], e[1]), pack_h2(e[2], e[3]), 0u, 0u};
            const f16x8_t pf = __builtin_bit_cast(f16x8_t, pk);
            const f32x4_t o0 = mfma_f16(v20, pf, z4), o1 = mfma_f16(v21, pf, z4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            *reinterpret_cast<u32x4_t*>(smem + G::L_XN + (h * 64 + lane) * 16) = pack_bf8(o0 * inv, o1 * inv);
          }
        } else {
        {
          DS_PHASE
          // q~[h] = q2[h] (16 queries x 32 d) x W_k1[h] (32 d x C channels): C / 16 output tiles of one MFMA each, dealt to the waves by (head, k-step)
#pragma unroll 1
          for (int u = wave; u < NH * KS; u += NW) {
            const int h = u / KS, ks = u - h * KS;
            const bf16x8_t q2f = *reinterpret_cast<const bf16x8_t*>(smem + G::L_H + (h * 64 + lane) * 16);
            const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4_t d0 = mfma_bf16(ld_frag(wp, G::WS_K1T + h * (C / 16) + 2 * ks, lane), q2f, z4);
            const f32x4_t d1 = mfma_bf16(ld_frag(wp, G::WS_K1T + h * (C / 16) + 2 * ks + 1, lane), q2f, z4);
            __builtin_amdgcn_raw_buffer_store_b128(pack_bf8(d0, d1), mr, ((h * G::MF_HEAD + 3 + ks) * 64 + lane) * 16, 0, 16);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32*)mflag, (unsigned)(gb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        DS_STAMP(11);
        asm volatile("; PHASE_MCOMB");
        // ---- the KWG partials of the c-direction attention -> c' (log-sum-exp combine, + v bias) -> proj_c operand in LDS ----
        if (wave0 == 0) {          // lane r polls the flag of image workgroup r
          unsigned spins = 0;
          while (true) {
            bool ok = true;
#pragma unroll
            for (int r0 = 0; r0 < KWG; r0 += 64) {
              const unsigned v = r0 + lane0 < KWG ? __hip_atomic_load((gu32*)(partflag + r0 + lane0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
              ok = ok && v >= (unsigned)(gb + 1);
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { if (lane0 == 0) __hip_atomic_store((gu32*)errflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          }
        }
        __syncthreads();
        DS_STAMP(12);
        {
          DS_PHASE
          const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)G::PART_BYTES, 0x00020000);
#pragma unroll 1
          for (int h = wave; h < NH; h += NW) {
            // (two passes over the KWG partials, CB workgroups per batch of loads: the loads of a batch are in flight together -- one at a time this was 28 x 2 dependent L2 round
            //  trips, 50 us per block at stage 1, and the whole image waited on it)
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
            constexpr int CB = KWG % 8 == 0 ? 8 : 7;
            static_assert(KWG % CB == 0, "batches of partials");
            float M = -INFINITY;
#pragma unroll 1
            for (int r0 = 0; r0 < KWG; r0 += CB) {
              float mb[CB];
#pragma unroll
              for (int k = 0; k < CB; ++k) mb[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pr, ((((r0 + k) * NH + h) * 3 + 2) * 64 + lane) * 16, 0, 16));
#pragma unroll
              for (int k = 0; k < CB; ++k) M = max2(M, mb[k]);
            }
            f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
            float L = 0.f;
#pragma unroll 1
            for (int r0 = 0; r0 < KWG; r0 += CB) {
              f32x4_t p0[CB], p1[CB]; u32x2_t ml[CB];
#pragma unroll
              for (int k = 0; k < CB; ++k) {
                const int base = (((r0 + k) * NH + h) * 3 * 64 + lane) * 16;
                p0[k] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(pr, base, 0, 16));
                p1[k] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(pr, base + 1024, 0, 16));
                ml[k] = __builtin_amdgcn_raw_buffer_load_b64(pr, base + 2048, 0, 16);
              }
#pragma unroll
              for (int k = 0; k < CB; ++k) {
                const float w = __builtin_amdgcn_exp2f(__uint_as_float(ml[k][0]) - M);
                L = fmaf(__uint_as_float(ml[k][1]), w, L);
                o0 += p0[k] * w; o1 += p1[k] * w;
              }
            }
            const float inv = 1.f / L;
            const float* bv = vec + G::V_QKV1B + 2 * C + 32 * h + 4 * g;
            const float4 b0 = *reinterpret_cast<const float4*>(bv), b1 = *reinterpret_cast<const float4*>(bv + 16);
            const f32x4_t c0 = {o0[0] * inv + b0.x, o0[1] * inv + b0.y, o0[2] * inv + b0.z, o0[3] * inv + b0.w};
            const f32x4_t c1 = {o1[0] * inv + b1.x, o1[1] * inv + b1.y, o1[2] * inv + b1.z, o1[3] * inv + b1.w};
            *reinterpret_cast<u32x4_t*>(smem + G::L_XN + (h * 64 + lane) * 16) = pack_bf8(c0, c1);
          }
        }
        }          // (KIND != 2)
        __syncthreads();
        DS_STAMP(13);
        asm volatile("; PHASE_MPROJ");
        // ---- c += proj_c(c') + bias; norm2; MLP ----
        {
          DS_PHASE
          float4 pb[CT];
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) pb[ct] = *reinterpret_cast<const float4*>(vec + G::V_PCB + CW * wave + 16 * ct + 4 * g);
          bf16x8_t ringp[G::PRD][CT];
          const unsigned char* wcur = wp + (size_t)(G::WS_PC + wave * (CT * KS)) * 1024;
          ring_fill<CT, G::PRD>(ringp, wcur, lane);
          gemm_unit<CT, KS, G::PRD, true, 1>(Rc, ringp, wcur, wcur, smem + G::L_XN, lane);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) { Rc[0][ct][0] += pb[ct].x; Rc[0][ct][1] += pb[ct].y; Rc[0][ct][2] += pb[ct].z; Rc[0][ct][3] += pb[ct].w; }
          __syncthreads();          // every wave has read the proj_c operand: norm2 may overwrite it
          ds_layer_norm<NW, 1, CT>(Rc, vec + G::V_N2W, vec + G::V_N2B, a.eps, smem + G::L_XN, stat, wave, lane);
        }
        DS_STAMP(14);
        ds_mlp<NW, 1, G>(Rc, wp, vec, smem + G::L_XN, smem + G::L_H, lane0, wave0);
        DS_STAMP(15);
        {
          DS_PHASE
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const float4 b = *reinterpret_cast<const float4*>(vec + G::V_FC2B + CW * wave + 16 * ct + 4 * g);
            Rc[0][ct][0] += b.x; Rc[0][ct][1] += b.y; Rc[0][ct][2] += b.z; Rc[0][ct][3] += b.w;
          }
        }
        __syncthreads();
      }
      {
        DS_PHASE
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float f[4] = {Rc[0][ct][0], Rc[0][ct][1], Rc[0][ct][2], Rc[0][ct][3]};
          st4(a.c_out + ((size_t)img * DS_M + li) * C + CW * wave + 16 * ct + 4 * g, f);
        }
      }
      continue;
    }

    asm volatile("; PHASE_IMG");
    // =================================== an image workgroup: 112 image tokens (ROWS grid rows) of image `img` ===================================
    f32x4_t R[NT][CT];
    auto load_rows = [&](int first_token) {
      DS_PHASE
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16_t* src = a.x_in + ((size_t)img * G::NIMG + first_token + 16 * t + li) * C;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { float f[4]; ld4(src + CW * wave + 16 * ct + 4 * g, f); R[t][ct] = f32x4_t{f[0], f[1], f[2], f[3]}; }
      }
    };
    if constexpr (KIND != 1) load_rows(rolep * TOK);
#pragma unroll 1
    for (int it = 0; it < a.nblocks * sub_n; ++it) {
      const int blk = sub_n == 1 ? it : it / sub_n, role = sub_n == 1 ? rolep : rolep * sub_n + (it - blk * sub_n), tok0 = role * TOK;      // role: the row group [role ROWS, (role + 1) ROWS)
      const int gb = round * a.nblocks + blk;
      const int lane = lane0, wave = wave0;      // (stamps only)
      const unsigned char* const wp = reinterpret_cast<const unsigned char*>(a.wpk) + (size_t)blk * G::WS_FRAGS * 1024;
      const float* const vec = a.vec + (size_t)blk * G::V_FLOATS;
      if constexpr (KIND == 1) load_rows(tok0);          // "C" blocks return x as it came: the position embedding of a block only feeds its norm1, every block (and row group) starts from x_in
      DS_STAMP(0);
      if (a.timing && gb == a.timing_block && lane0 == 0) {          // (placement probe: which CU / SE / XCD hosts this workgroup -- tools/dstage_timeline.py)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.timing[((size_t)blockIdx.x * NW + wave0) * DS_NSTAMP + 15] = ((unsigned long long)xcc << 32) | hw;
      }
      asm volatile("; PHASE_DW");
      // ---- x += dwconv3x3(x) + bias: per channel tile a wave-private bf16 image [ROWS + 2][GW + 2] of its 16 channels, zero pads, rows across the cuts from the peers ----
      {
        DS_PHASE
        unsigned char* const stg = smem + wave * G::STG_WAVE;
        const unsigned char* const hprev = halo + (size_t)((gb + 1) & 1) * (G::HALO_BYTES / 2);      // the rows published at the end of block gb - 1
        float4 wq[2][10];          // (requested in front of the halo wait: csrc/sstage.hip)
#pragma unroll
        for (int e = 0; e < 9; ++e) wq[0][e] = *reinterpret_cast<const float4*>(vec + G::V_POSW + (CW * wave + 4 * g) * 9 + 4 * e);
        wq[0][9] = *reinterpret_cast<const float4*>(vec + G::V_POSB + CW * wave + 4 * g);
        if (gb > 0 && KIND != 1) {
          // (block 0 of a later image reads its halo from x_in, but still waits: a workgroup must not run two blocks ahead of a neighbour that reads its rows)
          if (role > 0) wait_flag(haloflag + role - 1, (unsigned)gb, errflag, lane);
          if (role + 1 < KWG) wait_flag(haloflag + role + 1, (unsigned)gb, errflag, lane);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          int l2 = lane; asm volatile("" : "+v"(l2));
          const int g = l2 >> 4, li = l2 & 15;
          const int c0 = CW * wave + 16 * ct;
          if (ct + 1 < CT) {
#pragma unroll
            for (int e = 0; e < 9; ++e) wq[(ct + 1) & 1][e] = *reinterpret_cast<const float4*>(vec + G::V_POSW + (c0 + 16 + 4 * g) * 9 + 4 * e);
            wq[(ct + 1) & 1][9] = *reinterpret_cast<const float4*>(vec + G::V_POSB + c0 + 16 + 4 * g);
          }
          // own rows
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int s = 16 * t + li, y = s / GW, x = s - y * GW;
            *reinterpret_cast<uint2*>(stg + ((y + 1) * G::STG_COLS + x + 1) * G::STG_E + 8 * g) = make_uint2(pack_bf2(R[t][ct][0], R[t][ct][1]), pack_bf2(R[t][ct][2], R[t][ct][3]));
          }
          // pad columns, and the rows beyond the image
          for (int e = l2; e < 2 * (ROWS + 2) * 2; e += 64) {          // (entry, 16-byte half)
            const int ent = e >> 1, row = ent >> 1, col = (ent & 1) * (GW + 1);
            *reinterpret_cast<uint2*>(stg + (row * G::STG_COLS + col) * G::STG_E + (e & 1) * 16) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(stg + (row * G::STG_COLS + col) * G::STG_E + (e & 1) * 16 + 8) = make_uint2(0u, 0u);
          }
          // the row above (last grid row of workgroup role - 1) and below (first grid row of role + 1): 2 x 16-byte pieces per token
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const int nb = side ? role + 1 : role - 1, srow = side ? ROWS + 1 : 0;
            const bool inside = side ? role + 1 < KWG : role > 0;
            for (int p = l2; p < GW * 2; p += 64) {
              const int tok = p >> 1, q = p & 1;
              u32x4_t hv = {0u, 0u, 0u, 0u};
              if (inside) {
                u32x4_t v;
                if (blk == 0 || KIND == 1) v = *reinterpret_cast<const u32x4_t*>(a.x_in + ((size_t)img * G::NIMG + (side ? tok0 + TOK : tok0 - GW) + tok) * C + c0 + 8 * q);
                else {
                  const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hprev), 0, (int)(G::HALO_BYTES / 2), 0x00020000);
                  v = __builtin_amdgcn_raw_buffer_load_b128(hr, (((nb * 2 + (side ? 0 : 1)) * GW + tok) * C + c0 + 8 * q) * 2, 0, 16);
                }
                hv = v;
              }
              *reinterpret_cast<uint2*>(stg + (srow * G::STG_COLS + tok + 1) * G::STG_E + 16 * q) = make_uint2(hv[0], hv[1]);
              *reinterpret_cast<uint2*>(stg + (srow * G::STG_COLS + tok + 1) * G::STG_E + 16 * q + 8) = make_uint2(hv[2], hv[3]);
            }
          }
          // tap weights: bf16 pairs (w, 0) / (0, w), packed by lmv_*stage_pack (stage_common.h): v_dot2c_f32_bf16 of a loaded channel pair with one of them is that channel's tap product,
          // accumulated in fp32 (bf16 taps, as the reference's autocast convolution; the fp16 image of an earlier version overflowed on residual streams beyond 65504)
          unsigned wt[36];
#pragma unroll
          for (int e = 0; e < 9; ++e) { const float4 v = wq[ct & 1][e]; wt[4 * e] = __float_as_uint(v.x); wt[4 * e + 1] = __float_as_uint(v.y); wt[4 * e + 2] = __float_as_uint(v.z); wt[4 * e + 3] = __float_as_uint(v.w); }
          const float4 pb = wq[ct & 1][9];
          // the 9 taps of token tile t + 1 are requested before the 36 dot products of tile t are issued (two statically named register sets: csrc/sstage.hip, round 5)
          uint2 f[2][9];
          auto tap_base = [&](int t) -> const unsigned char* {
            const int s = 16 * t + li, y = s / GW, x = s - y * GW;
            return stg + (y * G::STG_COLS + x) * G::STG_E + 8 * g;      // entry of the (-1, -1) neighbour
          };
          {
            const unsigned char* const tap0 = tap_base(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) f[0][tap] = *reinterpret_cast<const uint2*>(tap0 + ((tap / 3) * G::STG_COLS + tap % 3) * G::STG_E);
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) {
              const unsigned char* const tap1 = tap_base(t + 1);
#pragma unroll
              for (int tap = 0; tap < 9; ++tap) f[(t + 1) & 1][tap] = *reinterpret_cast<const uint2*>(tap1 + ((tap / 3) * G::STG_COLS + tap % 3) * G::STG_E);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) asm volatile("" : "+v"(f[t & 1][tap]));
            float acc[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
              acc[0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, f[t & 1][tap].x), __builtin_bit_cast(bf16x2_t, wt[0 * 9 + tap]), acc[0], false);
              acc[1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, f[t & 1][tap].x), __builtin_bit_cast(bf16x2_t, wt[1 * 9 + tap]), acc[1], false);
              acc[2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, f[t & 1][tap].y), __builtin_bit_cast(bf16x2_t, wt[2 * 9 + tap]), acc[2], false);
              acc[3] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, f[t & 1][tap].y), __builtin_bit_cast(bf16x2_t, wt[3 * 9 + tap]), acc[3], false);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) R[t][ct][r] += acc[r];
            asm volatile("" : "+v"(R[t][ct]));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      DS_STAMP(1);
      asm volatile("; PHASE_LN1");
      // ---- norm1 -> LDS ----
      {
        DS_PHASE
        ds_layer_norm<NW, NT, CT>(R, vec + G::V_N1W, vec + G::V_N1B, a.eps, smem + G::L_XN, stat, wave, lane);
        if constexpr (KIND != 2)
          if (wave == 0) wait_flag(mflag, (unsigned)(gb + 1), errflag, lane);      // the meta workgroup's fragments of this block
      }
      __syncthreads();
      DS_STAMP(2);
      if constexpr (KIND != 2) {
      const unsigned char* const mf = mfr + (size_t)(gb & 1) * (G::MFRAG_BYTES / 2);
      const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(mf), 0, (int)(G::MFRAG_BYTES / 2), 0x00020000);
      const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)G::PART_BYTES, 0x00020000);
      asm volatile("; PHASE_CDIR");
      // ---- c-direction: per head, scores of the 16 meta queries against the 112 tokens of this workgroup (q~ . norm1(x)), local softmax, v1 of the head
      //      (operands swapped: the V^T fragments of P V), partial (max, sum, sum p v) -> the meta workgroup ----
      {
        DS_PHASE
        const int nc = (wave >= NW / 2 && NW + wave - NW / 2 < NH) ? 2 : 1;
#pragma unroll 1
        for (int hu = 0; hu < nc; ++hu) {
          const int h = hu == 0 ? wave : NW + wave - NW / 2;
          f32x4_t S[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) S[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t qt = as_bf8(__builtin_amdgcn_raw_buffer_load_b128(mr, ((h * G::MF_HEAD + 3 + ks) * 64 + lane) * 16, 0, 16));
#pragma unroll
            for (int t = 0; t < NT; ++t) S[t] = mfma_bf16(*reinterpret_cast<const bf16x8_t*>(smem + G::L_XN + ((ks * NT + t) * 64 + lane) * 16), qt, S[t]);
          }
          float m = -INFINITY;
#pragma unroll
          for (int t = 0; t < NT; ++t) m = max2(m, max4(S[t]));
          m = xmax4(m);
          float l = 0.f;
          u32x4_t P[NP];          // pairs of key tiles (the last pair of an odd tile count: one tile, zeros)
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            constexpr int LAST = NT - 1;
            const bool two = 2 * p + 1 < NT;
            float e[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(S[2 * p][r] - m); e[4 + r] = two ? __builtin_amdgcn_exp2f(S[two ? 2 * p + 1 : LAST][r] - m) : 0.f; }
            l += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
            P[p] = u32x4_t{pack_h2(e[0], e[1]), pack_h2(e[2], e[3]), pack_h2(e[4], e[5]), pack_h2(e[6], e[7])};
          }
          l = xsum4(l);
          asm volatile("" : "+v"(l));
          // v1 of the head: D'[token][channel] (no bias: sum p = 1 after the combine, the meta workgroup adds it)
          f32x4_t acc[NT][2];
#pragma unroll
          for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
          bf16x8_t ring[URD][2];
          const unsigned char* wcur = wp + (size_t)(G::WS_V1 + h * (2 * KS)) * 1024;
          ring_fill<2, URD>(ring, wcur, lane);
          gemm_unit<2, KS, URD, false, NT>(acc, ring, wcur, wcur, smem + G::L_XN, lane);
          f32x4_t O[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              const f32x4_t lo = acc[2 * p][dt];
              u32x4_t vk = {pack_h2(lo[0], lo[1]), pack_h2(lo[2], lo[3]), 0u, 0u};
              if (2 * p + 1 < NT) { const f32x4_t hi = acc[2 * p + 1 < NT ? 2 * p + 1 : 0][dt]; vk[2] = pack_h2(hi[0], hi[1]); vk[3] = pack_h2(hi[2], hi[3]); }
              O[dt] = mfma_f16(__builtin_bit_cast(f16x8_t, vk), __builtin_bit_cast(f16x8_t, P[p]), O[dt]);
            }
          const int pbase = ((role * NH + h) * 3 * 64 + lane) * 16;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, O[0]), pr, pbase, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, O[1]), pr, pbase + 1024, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(m), __float_as_uint(l), 0u, 0u}, pr, pbase + 2048, 0, 16);
        }
      }
      DS_STAMP(3);
      if constexpr (KIND == 1) {          // "C" blocks end here for the image tokens
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32*)(partflag + role), (unsigned)(gb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      asm volatile("; PHASE_XDIR");
      // ---- x-direction: per head, q1 of the 112 tokens, softmax over the 16 meta keys, P V2; the proj_x operand fragments wait in registers ----
      const int nx = wave0 + NW < NH ? 2 : 1;
      u32x4_t AO1[NT];          // the second head of waves 0 .. NW / 2 - 1
      {
        DS_PHASE
        bf16x8_t ring[URD][2];
#pragma unroll
        for (int hu = 0; hu < 2; ++hu) {
          if (hu < nx) {
            const int h = wave + NW * hu;
            const u32x4_t k2 = __builtin_amdgcn_raw_buffer_load_b128(mr, ((h * G::MF_HEAD) * 64 + lane) * 16, 0, 16);
            const u32x4_t v20 = __builtin_amdgcn_raw_buffer_load_b128(mr, ((h * G::MF_HEAD + 1) * 64 + lane) * 16, 0, 16);
            const u32x4_t v21 = __builtin_amdgcn_raw_buffer_load_b128(mr, ((h * G::MF_HEAD + 2) * 64 + lane) * 16, 0, 16);
            const unsigned char* wcur = wp + (size_t)(G::WS_Q1 + h * (2 * KS)) * 1024;
            ring_fill<2, URD>(ring, wcur, lane);
            const float* bq = vec + G::V_QKV1B + 32 * h + 4 * g;
            const float4 b0 = *reinterpret_cast<const float4*>(bq), b1 = *reinterpret_cast<const float4*>(bq + 16);
            f32x4_t acc[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
            gemm_unit<2, KS, URD, true, NT>(acc, ring, wcur, wcur, smem + G::L_XN, lane);
            const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const f32x4_t q0 = {(acc[t][0][0] + b0.x) * a.sx, (acc[t][0][1] + b0.y) * a.sx, (acc[t][0][2] + b0.z) * a.sx, (acc[t][0][3] + b0.w) * a.sx};
              const f32x4_t q1 = {(acc[t][1][0] + b1.x) * a.sx, (acc[t][1][1] + b1.y) * a.sx, (acc[t][1][2] + b1.z) * a.sx, (acc[t][1][3] + b1.w) * a.sx};
              const f32x4_t s = mfma_bf16(as_bf8(k2), as_bf8(pack_bf8(q0, q1)), z4);          // S^T[key][query]
              const float m = xmax4(max4(s));
              float e[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(s[r] - m);
              const float inv = 1.f / xsum4((e[0] + e[1]) + (e[2] + e[3]));
              const u32x4_t pk = {pack_h2(e[0], e[1]), pack_h2(e[2], e[3]), 0u, 0u};
              const f16x8_t pf = __builtin_bit_cast(f16x8_t, pk);
              const f32x4_t o0 = mfma_f16(__builtin_bit_cast(f16x8_t, v20), pf, z4), o1 = mfma_f16(__builtin_bit_cast(f16x8_t, v21), pf, z4);
              const u32x4_t ao = pack_bf8(o0 * inv, o1 * inv);
              if (hu == 0) *reinterpret_cast<u32x4_t*>(smem + G::L_AO + ((h * NT + t) * 64 + lane) * 16) = ao;
              else AO1[t] = ao;
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the partial stores of this wave (R1: every storing wave drains)
      __syncthreads();                                         // ... and every wave is done with the LayerNorm output in LDS
      if (tid == 0) __hip_atomic_store((gu32*)(partflag + role), (unsigned)(gb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nx == 2) {
        DS_PHASE
        const int h = wave + NW;
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<u32x4_t*>(smem + G::L_AO + ((h * NT + t) * 64 + lane) * 16) = AO1[t];
      }
      } else {
        // =========== "S" blocks: self-attention of the image tokens (models/lemevit.py:185-205), the keys and values of the whole image through L2 ===========
        constexpr int KT = KWG * NT, NPW = NT / 2, NQ = NT / 2;          // key tiles of the image; key-tile pairs / query tiles of a unit per workgroup
#ifndef DS_RG1
#define DS_RG1 3
#endif
#ifndef DS_RG2
#define DS_RG2 3
#endif
        constexpr int RG1 = DS_RG1, RG2 = DS_RG2;          // key-tile pairs in flight (ring depth) of the two attention passes
        static_assert(GW == 24 && NT % 2 == 0 && (2 * NH) % NW == 0 && NW % 2 == 0, "S blocks: the 24 x 24 instances");
        unsigned char* const kvw = kvb + (size_t)(gb & 1) * (2 * G::KV_HALF);
        const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(kvw, 0, (int)(2 * G::KV_HALF), 0x00020000);
        // ---- k (even waves) / v (odd waves) of this workgroup's tokens, 2 NH units over the waves -> K fragments [head][key tile], V^T fragments [head][pair][d-tile] ----
        {
          DS_PHASE
          bf16x8_t ring[URD][2];
          const int isv = wave & 1;
#pragma unroll 1
          for (int j = 0; j < (2 * NH) / NW; ++j) {
            const int h = (wave >> 1) + (NW / 2) * j;
            const unsigned char* wcur = wp + (size_t)((isv ? G::WS_V1 : G::WS_K1) + h * (2 * KS)) * 1024;
            ring_fill<2, URD>(ring, wcur, lane);
            f32x4_t acc[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
            if (!isv) {
              const float* bk = vec + G::V_QKV1B + C + 32 * h + 4 * g;
              const float4 b0 = *reinterpret_cast<const float4*>(bk), b1 = *reinterpret_cast<const float4*>(bk + 16);
              gemm_unit<2, KS, URD, true, NT>(acc, ring, wcur, wcur, smem + G::L_XN, lane);
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const f32x4_t k0 = {acc[t][0][0] + b0.x, acc[t][0][1] + b0.y, acc[t][0][2] + b0.z, acc[t][0][3] + b0.w};
                const f32x4_t k1 = {acc[t][1][0] + b1.x, acc[t][1][1] + b1.y, acc[t][1][2] + b1.z, acc[t][1][3] + b1.w};
                __builtin_amdgcn_raw_buffer_store_b128(pack_bf8(k0, k1), kr, ((h * KT + role * NT + t) * 64 + lane) * 16, 0, 16);
              }
            } else {
              const float bv[2] = {vec[G::V_QKV1B + 2 * C + 32 * h + li], vec[G::V_QKV1B + 2 * C + 32 * h + 16 + li]};
              gemm_unit<2, KS, URD, false, NT>(acc, ring, wcur, wcur, smem + G::L_XN, lane);
#pragma unroll
              for (int p = 0; p < NPW; ++p)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                  const f32x4_t lo = acc[2 * p][dt] + bv[dt], hi = acc[2 * p + 1][dt] + bv[dt];
                  const u32x4_t vk = {pack_h2(lo[0], lo[1]), pack_h2(lo[2], lo[3]), pack_h2(hi[0], hi[1]), pack_h2(hi[2], hi[3])};
                  __builtin_amdgcn_raw_buffer_store_b128(vk, kr, (int)G::KV_HALF + (((h * (KT / 2) + role * NPW + p) * 2 + dt) * 64 + lane) * 16, 0, 16);
                }
            }
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32*)(kvflag + role), (unsigned)(gb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        DS_STAMP(3);
        if (wave0 == 0) {          // every workgroup of the image has published its keys and values of this block
          unsigned spins = 0;
          while (true) {
            const unsigned v = lane0 < KWG ? __hip_atomic_load((gu32*)(kvflag + lane0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
            if (__all(v >= (unsigned)(gb + 1))) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { if (lane0 == 0) __hip_atomic_store((gu32*)errflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          }
        }
        __syncthreads();
        // ---- attention: 2 NH units (head, half of the workgroup's query tiles) over the waves; two passes over the KT key tiles (row maxima; exp2 / sums / P V with the scores
        //      recomputed), K and V^T fragments straight from L2 through a 3-deep register ring of key-tile pairs ----
#pragma unroll
        for (int j = 0; j < (2 * NH) / NW; ++j) {
          DS_PHASE
          const int h = (wave >> 1) + (NW / 2) * j, t0 = NQ * (wave & 1);
          bf16x8_t Q[NQ];
          {
            bf16x8_t ring[URD][2];
            const unsigned char* wcur = wp + (size_t)(G::WS_Q1 + h * (2 * KS)) * 1024;
            ring_fill<2, URD>(ring, wcur, lane);
            const float* bq = vec + G::V_QKV1B + 32 * h + 4 * g;
            const float4 b0 = *reinterpret_cast<const float4*>(bq), b1 = *reinterpret_cast<const float4*>(bq + 16);
            f32x4_t acc[NQ][2];
#pragma unroll
            for (int t = 0; t < NQ; ++t) { acc[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
            gemm_unit<2, KS, URD, true, NQ, NT>(acc, ring, wcur, wcur, smem + G::L_XN + t0 * 1024, lane);
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
              const f32x4_t q0 = {(acc[t][0][0] + b0.x) * a.sx, (acc[t][0][1] + b0.y) * a.sx, (acc[t][0][2] + b0.z) * a.sx, (acc[t][0][3] + b0.w) * a.sx};
              const f32x4_t q1 = {(acc[t][1][0] + b1.x) * a.sx, (acc[t][1][1] + b1.y) * a.sx, (acc[t][1][2] + b1.z) * a.sx, (acc[t][1][3] + b1.w) * a.sx};
              Q[t] = as_bf8(pack_bf8(q0, q1));
            }
          }
          // (the attention outputs of heads >= NW land where the LayerNorm output lies: every wave must be done with its q projections first)
          if (NW / 2 * j + NW / 2 > NW) __syncthreads();
          const int kbase = (h * KT * 64 + lane) * 16, vbase = (int)G::KV_HALF + (h * KT * 64 + lane) * 16;
          const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
          float m[NQ];
#pragma unroll
          for (int t = 0; t < NQ; ++t) m[t] = -INFINITY;
          {
            u32x4_t kp[RG1][2];
#pragma unroll
            for (int pi = 0; pi < RG1 - 1; ++pi)
#pragma unroll
              for (int e = 0; e < 2; ++e) kp[pi][e] = __builtin_amdgcn_raw_buffer_load_b128(kr, kbase + (2 * pi + e) * 1024, 0, 16);
#pragma unroll
            for (int pi = 0; pi < KT / 2; ++pi) {
              if (pi + RG1 - 1 < KT / 2) {
#pragma unroll
                for (int e = 0; e < 2; ++e) kp[(pi + RG1 - 1) % RG1][e] = __builtin_amdgcn_raw_buffer_load_b128(kr, kbase + (2 * (pi + RG1 - 1) + e) * 1024, 0, 16);
              }
#pragma unroll
              for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int t = 0; t < NQ; ++t) { const f32x4_t sc = mfma_bf16(as_bf8(kp[pi % RG1][e]), Q[t], z4); m[t] = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), m[t]); }
            }
          }
          // second pass: the row maximum rides the score MFMA as its accumulator input (s - m for free), the row sums come out of one more MFMA against a fragment of ones
          // (sum over the 32 keys of the pair AND over the lane groups: no adds, no cross-lane step) -- the pass is VALU-bound, the matrix pipe has the room
          f32x4_t L[NQ], O[NQ][2], negm[NQ];
          const f16x8_t ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
#pragma unroll
          for (int t = 0; t < NQ; ++t) { m[t] = xmax4(m[t]); negm[t] = f32x4_t{-m[t], -m[t], -m[t], -m[t]}; L[t] = z4; O[t][0] = z4; O[t][1] = z4; asm volatile("" : "+v"(Q[t])); }
          {
            int ko = 0; asm volatile("" : "+v"(ko));          // (an opaque offset: the second pass's loads and score tiles are not the first pass's)
            u32x4_t kp[RG2][2], vp[RG2][2];
#pragma unroll
            for (int pi = 0; pi < RG2 - 1; ++pi)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                kp[pi][e] = __builtin_amdgcn_raw_buffer_load_b128(kr, ko + kbase + (2 * pi + e) * 1024, 0, 16);
                vp[pi][e] = __builtin_amdgcn_raw_buffer_load_b128(kr, ko + vbase + (2 * pi + e) * 1024, 0, 16);
              }
#pragma unroll
            for (int pi = 0; pi < KT / 2; ++pi) {
              if (pi + RG2 - 1 < KT / 2) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  kp[(pi + RG2 - 1) % RG2][e] = __builtin_amdgcn_raw_buffer_load_b128(kr, ko + kbase + (2 * (pi + RG2 - 1) + e) * 1024, 0, 16);
                  vp[(pi + RG2 - 1) % RG2][e] = __builtin_amdgcn_raw_buffer_load_b128(kr, ko + vbase + (2 * (pi + RG2 - 1) + e) * 1024, 0, 16);
                }
              }
#pragma unroll
              for (int t = 0; t < NQ; ++t) {
                const f32x4_t s0 = mfma_bf16(as_bf8(kp[pi % RG2][0]), Q[t], negm[t]), s1 = mfma_bf16(as_bf8(kp[pi % RG2][1]), Q[t], negm[t]);
                float e[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(s0[r]); e[4 + r] = __builtin_amdgcn_exp2f(s1[r]); }
                const u32x4_t pk = {pack_h2(e[0], e[1]), pack_h2(e[2], e[3]), pack_h2(e[4], e[5]), pack_h2(e[6], e[7])};
                const f16x8_t pf = __builtin_bit_cast(f16x8_t, pk);
                O[t][0] = mfma_f16(__builtin_bit_cast(f16x8_t, vp[pi % RG2][0]), pf, O[t][0]);
                O[t][1] = mfma_f16(__builtin_bit_cast(f16x8_t, vp[pi % RG2][1]), pf, O[t][1]);
                L[t] = mfma_f16(ones, pf, L[t]);
              }
            }
          }
#pragma unroll
          for (int t = 0; t < NQ; ++t) {
            const float inv = 1.f / L[t][0];
            *reinterpret_cast<u32x4_t*>(smem + G::L_AO + ((h * NT + t0 + t) * 64 + lane) * 16) = pack_bf8(O[t][0] * inv, O[t][1] * inv);
          }
        }
      }
      __syncthreads();
      DS_STAMP(4);
      asm volatile("; PHASE_PROJ");
      // ---- x += proj_x(attention) + bias; norm2; MLP ----
      {
        DS_PHASE
        float4 pb[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) pb[ct] = *reinterpret_cast<const float4*>(vec + G::V_PXB + CW * wave + 16 * ct + 4 * g);
        bf16x8_t ringp[G::PRD][CT];
        const unsigned char* wcur = wp + (size_t)(G::WS_PX + wave * (CT * KS)) * 1024;
        ring_fill<CT, G::PRD>(ringp, wcur, lane);
        gemm_unit<CT, KS, G::PRD, true, NT>(R, ringp, wcur, wcur, smem + G::L_AO, lane);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int t = 0; t < NT; ++t) { R[t][ct][0] += pb[ct].x; R[t][ct][1] += pb[ct].y; R[t][ct][2] += pb[ct].z; R[t][ct][3] += pb[ct].w; }
        __syncthreads();          // every wave has read the proj_x operand: norm2 may overwrite it
        ds_layer_norm<NW, NT, CT>(R, vec + G::V_N2W, vec + G::V_N2B, a.eps, smem + G::L_XN, stat, wave, lane);
      }
      DS_STAMP(5);
      asm volatile("; PHASE_MLP");
      ds_mlp<NW, NT, G>(R, wp, vec, smem + G::L_XN, smem + G::L_H, lane0, wave0);
      DS_STAMP(6);
      asm volatile("; PHASE_END");
      // ---- block end: + mlp.3.bias; the first and the last grid row go to the neighbours ----
      {
        DS_PHASE
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float4 b = *reinterpret_cast<const float4*>(vec + G::V_FC2B + CW * wave + 16 * ct + 4 * g);
#pragma unroll
          for (int t = 0; t < NT; ++t) { R[t][ct][0] += b.x; R[t][ct][1] += b.y; R[t][ct][2] += b.z; R[t][ct][3] += b.w; }
        }
        unsigned char* const hcur = halo + (size_t)(gb & 1) * (G::HALO_BYTES / 2);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int s = 16 * t + li;
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const int tok = side ? s - (TOK - GW) : s;
            if ((unsigned)tok < (unsigned)GW) {
              bf16_t* dst = reinterpret_cast<bf16_t*>(hcur) + ((size_t)(role * 2 + side) * GW + tok) * C + CW * wave + 4 * g;
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                const unsigned long long pk = (unsigned long long)pack_bf2(R[t][ct][0], R[t][ct][1]) | ((unsigned long long)pack_bf2(R[t][ct][2], R[t][ct][3]) << 32);
                __hip_atomic_store((gu64*)(dst + 16 * ct), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_store((gu32*)(haloflag + role), (unsigned)(gb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      DS_STAMP(7);
    }
    if constexpr (KIND != 1)
    {
      DS_PHASE
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bf16_t* dst = a.x_out + ((size_t)img * G::NIMG + rolep * TOK + 16 * t + li) * C;          // ("D" blocks: role = rolep)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float f[4] = {R[t][ct][0], R[t][ct][1], R[t][ct][2], R[t][ct][3]};
          st4(dst + CW * wave + 16 * ct + 4 * g, f);
        }
      }
    }
  }
}

// ---- packing ----------------------------------------------------------------------------------------------------------------------
struct DPackArgs { const bf16_t* qkv1_w; const bf16_t* qkv2_w; const bf16_t* projx_w; const bf16_t* projc_w; const bf16_t* fc1_w; const bf16_t* fc2_w; uint4* out; };

template <int NW, int CT>
__global__ __launch_bounds__(256) void dstage_pack_kernel(const DPackArgs a) {
  using G = DG<NW, NW == 8 ? 24 : NW == 4 ? 28 : 56, CT>;          // (the packed layout does not depend on the grid)
  constexpr int C = G::C, KS = G::KS, KSC = G::KSC, UF = 2 * KS, PF = CT * KS, F2 = CT * KSC, CW = G::CW;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  if (f >= G::WS_FRAGS) return;
  const bf16_t* base; int row0, ks, ld = C;
  if (f >= G::WS_K1T && f < G::WS_PC) {
    // transposed fragment of the k rows of qkv1: lane (g, i) holds W[C + 32 h + 16 (j >> 2) + 4 g + (j & 3)][16 m + i]
    const int q = f - G::WS_K1T, h = q / (C / 16), m = q - h * (C / 16);
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.qkv1_w[(size_t)(C + 32 * h + 16 * (j >> 2) + 4 * g + (j & 3)) * C + 16 * m + i];
    a.out[(size_t)f * 64 + lane] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    return;
  }
  if (f < G::WS_V1) { const int h = f / UF, r = f - h * UF; ks = r >> 1; base = a.qkv1_w; row0 = 32 * h + 16 * (r & 1); }
  else if (f < G::WS_PX) { const int q = f - G::WS_V1, h = q / UF, r = q - h * UF; ks = r >> 1; base = a.qkv1_w; row0 = 2 * C + 32 * h + 16 * (r & 1); }
  else if (f < G::WS_FC1) { const int q = f - G::WS_PX, w = q / PF, r = q - w * PF; ks = r / CT; base = a.projx_w; row0 = CW * w + 16 * (r - ks * CT); }
  else if (f < G::WS_FC2) { const int q = f - G::WS_FC1, cw = q / UF, r = q - cw * UF; ks = r >> 1; base = a.fc1_w; row0 = 32 * NW * (cw / NW) + 32 * (cw % NW) + 16 * (r & 1); }
  else if (f < G::WS_K2) { const int q = f - G::WS_FC2, cw = q / F2, r = q - cw * F2, ksl = r / CT; ks = KSC * (cw / NW) + ksl; base = a.fc2_w; ld = 4 * C; row0 = CW * (cw % NW) + 16 * (r - ksl * CT); }
  else if (f < G::WS_V2) { const int q = f - G::WS_K2, h = q / UF, r = q - h * UF; ks = r >> 1; base = a.qkv2_w; row0 = C + 32 * h + 16 * (r & 1); }
  else if (f < G::WS_Q2) { const int q = f - G::WS_V2, h = q / UF, r = q - h * UF; ks = r >> 1; base = a.qkv2_w; row0 = 2 * C + 32 * h + 16 * (r & 1); }
  else if (f < G::WS_K1T) { const int q = f - G::WS_Q2, h = q / UF, r = q - h * UF; ks = r >> 1; base = a.qkv2_w; row0 = 32 * h + 16 * (r & 1); }
  else if (f < G::WS_K1) { const int q = f - G::WS_PC, w = q / PF, r = q - w * PF; ks = r / CT; base = a.projc_w; row0 = CW * w + 16 * (r - ks * CT); }
  else { const int q = f - G::WS_K1, h = q / UF, r = q - h * UF; ks = r >> 1; base = a.qkv1_w; row0 = C + 32 * h + 16 * (r & 1); }
  const bf16_t* src = base + (size_t)(row0 + i) * ld + 32 * ks + 4 * g;
  const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
  a.out[(size_t)f * 64 + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// Workgroups of an instance the device holds at once: the occupancy query x the CU count (cached per device); without a device (sizing calls in a CPU process) the MI355X figure.
// Slots are assigned by ticket (stage_ticket), so the slot count below decides how many images are in flight, not whether the launch is correct; what progress needs -- room for
// 8 incomplete slots + 1 workgroup per launch in flight -- is what lmv_dstage_supported / lmv_dstage_max_concurrent check.
template <int NW, int GW, int CT, int KIND> static int ds_capacity_kind() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int v = cache[dev & 63].load(std::memory_order_relaxed);
  if (!v) { v = lmv_stage_capacity(reinterpret_cast<const void*>(dstage_kernel<NW, GW, CT, KIND>), 64 * NW, DG<NW, GW, CT>::L_TOTAL); cache[dev & 63].store(v, std::memory_order_relaxed); }
  return v;
}
template <int NW, int GW, int CT> static int ds_capacity(int kind) {
  int v;
  if constexpr (GW == 24) v = ds_capacity_kind<NW, GW, CT, 2>();
  else if constexpr (GW == 56 || GW == 96) v = kind == 1 ? ds_capacity_kind<NW, GW, CT, 1>() : ds_capacity_kind<NW, GW, CT, 0>();
  else v = ds_capacity_kind<NW, GW, CT, 0>();
  return v > 0 ? v : 256 * (8 / NW);
}
template <int NW, int GW, int CT> static int ds_nwg(int kind) { return kind == 1 ? DG<NW, GW, CT>::NWG_C : DG<NW, GW, CT>::NWG; }
template <int NW, int GW, int CT> static int ds_slots(int B, int kind) {
  int n = ds_capacity<NW, GW, CT>(kind) / ds_nwg<NW, GW, CT>(kind) / 8 * 8;          // whole groups of 8 slots (one per ticket counter)
  if (n < 8) n = 8;
  const int need = (B + 7) / 8 * 8;
  return n < need ? n : need;
}
template <int NW, int GW, int CT> static size_t ds_flag_bytes(int ns) { return (((size_t)ns * DG<NW, GW, CT>::FLAGS_PER_SLOT + 8 + 1) * 4 + 1023) / 1024 * 1024; }          // (+ the 8 ticket counters)
template <int NW, int GW, int CT> static size_t ds_workspace(int B) {          // (either kind)
  const int n0 = ds_slots<NW, GW, CT>(B, 0), n1 = ds_slots<NW, GW, CT>(B, 1), ns = n0 > n1 ? n0 : n1;
  return ds_flag_bytes<NW, GW, CT>(ns) + (size_t)ns * DG<NW, GW, CT>::SLOT_BYTES;
}

template <int NW, int GW, int CT> static int ds_pack_launch(const DPackArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((dstage_pack_kernel<NW, CT>), dim3((DG<NW, GW, CT>::WS_FRAGS + 3) / 4), dim3(256), 0, st, a);
  return 0;
}

template <int NW, int GW, int CT, int KIND> static int ds_launch_kind(const lmv_dstage_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* workspace, hipStream_t st) {
  using G = DG<NW, GW, CT>;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dstage_kernel<NW, GW, CT, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, G::L_TOTAL) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "dstage_fwd: cannot reserve LDS");
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int ns = ds_slots<NW, GW, CT>(d->B, KIND);
  const size_t flags = ds_flag_bytes<NW, GW, CT>(ns);
  unsigned char* ws = (unsigned char*)workspace;
  if (hipMemsetAsync(ws, 0, flags, st) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "dstage_fwd: flag reset failed");
  DsArgs a{};
  a.x_in = (const bf16_t*)x; a.c_in = (const bf16_t*)c; a.x_out = (bf16_t*)x_out; a.c_out = (bf16_t*)c_out;
  a.wpk = (const uint4*)d->wpk; a.vec = d->vec; a.flags = (unsigned*)ws; a.slots = ws + flags;
  a.B = d->B; a.nblocks = d->nblocks; a.nslots = ns; a.eps = d->eps;
  const double N = (double)d->H * d->W, lg2e = 1.4426950408889634;
  a.sx = (float)(log((double)d->M) / log(N) / sqrt((double)d->C) * lg2e);      // models/lemevit.py:255: log_N(M) C^-1/2
  a.sc = (float)(1.0 / sqrt(d->kind ? 32.0 : (double)d->C) * lg2e);            // :256; "C" blocks: F.scaled_dot_product_attention's head_dim^-1/2 (:480-483)
  if (KIND == 2) a.sx = a.sc;                                                  // "S" blocks: both attentions are F.scaled_dot_product_attention (:199-203)
  a.timing = (unsigned long long*)d->timing; a.timing_block = d->timing_block;
  a.err = lmv_stage_errword(st);
  if (!a.err) LMV_FAIL(LMV_ERR_LAUNCH, "dstage_fwd: cannot allocate the error word (the first stage call of a process must not be inside a stream capture)");
  a.tickets = a.flags + (size_t)ns * G::FLAGS_PER_SLOT; a.quota = (unsigned)(ns / 8 * (KIND == 1 ? G::NWG_C : G::NWG)); a.skew = (unsigned)lmv_config().stage_ticket_skew;
  hipLaunchKernelGGL((dstage_kernel<NW, GW, CT, KIND>), dim3(ns * (KIND == 1 ? G::NWG_C : G::NWG)), dim3(64 * NW), G::L_TOTAL, st, a);
  LMV_CHECK_LAUNCH("dstage_fwd");
  return LMV_OK;
}
template <int NW, int GW, int CT> static int ds_launch(const lmv_dstage_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* workspace, hipStream_t st) {
  if constexpr (GW == 24) {          // the "S"-block instances (stage 3 of LeMeViT-Base / -Tiny at 384 x 384)
    if (d->kind != 2) LMV_FAIL(LMV_ERR_DTYPE, "dstage_fwd: this shape is built for S blocks (kind 2) only");
    return ds_launch_kind<NW, GW, CT, 2>(d, x, c, x_out, c_out, workspace, st);
  } else {
    if (d->kind == 2) LMV_FAIL(LMV_ERR_DTYPE, "dstage_fwd: S blocks (kind 2) are built for the 24 x 24 grid (C = 384 / 192) only");
    if constexpr (GW == 56 || GW == 96) {          // "C" blocks exist at stage 0 only (the resolution of stage 1)
      if (d->kind) return ds_launch_kind<NW, GW, CT, 1>(d, x, c, x_out, c_out, workspace, st);
    } else if (d->kind) LMV_FAIL(LMV_ERR_DTYPE, "dstage_fwd: C blocks are built for the stage-0 / stage-1 grid only");
    return ds_launch_kind<NW, GW, CT, 0>(d, x, c, x_out, c_out, workspace, st);
  }
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------------------------------------
// the instances: 1000 NW + 10 GW + CT.  224 x 224 images: (4, 28, 3) / (2, 56, 3) LeMeViT-Base and -Small, (4, 28, 2) / (2, 56, 2) LeMeViT-Tiny; 384 x 384 (BASELINE config 5): (4, 48, 3) / (2, 96, 3), Tiny: (4, 48, 2) / (2, 96, 2)
#define DS_DISPATCH(code, EXPR, DFLT)                                                                                                                   \
  ((code) == 4283 ? EXPR(4, 28, 3) : (code) == 2563 ? EXPR(2, 56, 3) : (code) == 4282 ? EXPR(4, 28, 2) : (code) == 2562 ? EXPR(2, 56, 2) : (code) == 4483 ? EXPR(4, 48, 3) : \
   (code) == 2963 ? EXPR(2, 96, 3) : (code) == 4482 ? EXPR(4, 48, 2) : (code) == 2962 ? EXPR(2, 96, 2) : (code) == 8243 ? EXPR(8, 24, 3) : (code) == 4243 ? EXPR(4, 24, 3) : (DFLT))
static int ds_code_of_c(int C) { return C == 192 ? 4283 : C == 96 ? 2563 : C == 128 ? 4282 : C == 64 ? 2562 : C == 384 ? 8243 : 0; }          // (what depends on C only: the packed layout)
static int ds_code(int C, int H) {
  if (C == 192) return H == 28 ? 4283 : H == 48 ? 4483 : H == 24 ? 4243 : 0;          // (24 x 24: "S" blocks, kind 2)
  if (C == 96) return H == 56 ? 2563 : H == 96 ? 2963 : 0;
  if (C == 128) return H == 28 ? 4282 : H == 48 ? 4482 : 0;
  if (C == 64) return H == 56 ? 2562 : H == 96 ? 2962 : 0;
  if (C == 384) return H == 24 ? 8243 : 0;          // ("S" blocks, kind 2)
  return 0;
}
static int ds_variant(int C, int heads, int hidden, int H, int W, int M) {      // 0: not supported
  if (M != DS_M || H != W || heads != C / 32 || hidden != 4 * C) return 0;
  return ds_code(C, H);
}
int lmv_dstage_max_concurrent(int C, int H, int kind);
int lmv_dstage_supported(int C, int heads, int hidden, int H, int W, int M, int dtype) {
  if (!(dtype == LMV_BF16 && ds_variant(C, heads, hidden, H, W, M) != 0)) return 0;
  return lmv_dstage_max_concurrent(C, H, H == 24 ? 2 : 0) >= 1;          // a device too small for 8 incomplete slots + 1 workgroup (a partition) takes the per-block schedule
}
#define DS_WPK(NW, GW, CT) ((size_t)DG<NW, GW, CT>::WS_FRAGS * 1024)
size_t lmv_dstage_wpk_bytes(int C, int hidden) { (void)hidden; const int code = ds_code_of_c(C); return DS_DISPATCH(code, DS_WPK, (size_t)0); }
size_t lmv_dstage_vec_floats(int C, int hidden) { (void)hidden; return (size_t)27 * C; }
#define DS_WS(NW, GW, CT) ds_workspace<NW, GW, CT>(B)
size_t lmv_dstage_workspace_bytes(int B, int C) {          // (any grid the kernel takes at this C)
  const int code = ds_code_of_c(C), code2 = C == 192 ? 4483 : C == 96 ? 2963 : C == 128 ? 4482 : C == 64 ? 2962 : 0;
  const size_t a = DS_DISPATCH(code, DS_WS, (size_t)0), b = DS_DISPATCH(code2, DS_WS, (size_t)0), c3 = C == 192 ? ds_workspace<4, 24, 3>(B) : 0;
  return a > b ? (a > c3 ? a : c3) : (b > c3 ? b : c3);
}

// How many lmv_dstage_fwd calls may be in flight on different streams of one device.  Every launch holds at most 8 incomplete slots (one per ticket counter: stage_ticket,
// stage_common.h) of NWG - 1 resident workgroups each; as long as one more workgroup fits beside the incomplete slots of ALL launches in flight, the next workgroup to start
// completes a slot: n * 8 (NWG - 1) + 1 <= capacity.  Launches of different shapes mix under the same rule (each shape's n bounds its share of the device by 1 / n).
// 0: not even one launch (lmv_dstage_supported then says no).
template <int NW, int GW, int CT> static int ds_max_concurrent(int kind) {
  const int cap = ds_capacity<NW, GW, CT>(kind), nwg = ds_nwg<NW, GW, CT>(kind);
  return (cap - 1) / (8 * (nwg - 1));
}
#define DS_MAXC(NW, GW, CT) ds_max_concurrent<NW, GW, CT>(kind)
int lmv_dstage_max_concurrent(int C, int H, int kind) { const int code = ds_code(C, H); return DS_DISPATCH(code, DS_MAXC, 0); }

int lmv_dstage_pack(const lmv_dstage_block_params* p, void* wpk_out, float* vec_out, void* stream) {
  if (!p || !wpk_out || !vec_out) LMV_FAIL(LMV_ERR_SHAPE, "dstage_pack: null argument");
  const int code = ds_code_of_c(p->C);
  if (!(code && p->heads == p->C / 32 && p->hidden == 4 * p->C))
    LMV_FAIL(LMV_ERR_DTYPE, "dstage_pack: C = %d / heads = %d / hidden = %d is not a supported stage", p->C, p->heads, p->hidden);
  const void* ptrs[] = {p->qkv1_w, p->qkv2_w, p->projx_w, p->projc_w, p->fc1_w, p->fc2_w, p->n1_w, p->n1_b, p->qkv1_b, p->qkv2_b, p->projx_b, p->projc_b, p->n2_w, p->n2_b, p->fc1_b, p->fc2_b,
                        p->pos_w, p->pos_b, wpk_out, vec_out};
  for (const void* q : ptrs) if (!q || !lmv_aligned16(q)) LMV_FAIL(LMV_ERR_SHAPE, "dstage_pack: null or misaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  DPackArgs a{(const bf16_t*)p->qkv1_w, (const bf16_t*)p->qkv2_w, (const bf16_t*)p->projx_w, (const bf16_t*)p->projc_w, (const bf16_t*)p->fc1_w, (const bf16_t*)p->fc2_w, (uint4*)wpk_out};
#define DS_PACK(NW, GW, CT) ds_pack_launch<NW, GW, CT>(a, st)
  (void)DS_DISPATCH(code, DS_PACK, 0);
  LMV_CHECK_LAUNCH("dstage_pack");
  const int C = p->C;
  const struct { const float* src; int off, n; } v[] = {{p->n1_w, 0, C}, {p->n1_b, C, C}, {p->qkv1_b, 2 * C, 3 * C}, {p->qkv2_b, 5 * C, 3 * C}, {p->projx_b, 8 * C, C}, {p->projc_b, 9 * C, C},
                                                        {p->n2_w, 10 * C, C}, {p->n2_b, 11 * C, C}, {p->fc1_b, 12 * C, 4 * C}, {p->fc2_b, 16 * C, C}, {p->pos_b, 26 * C, C}};
  for (const auto& e : v)
    if (hipMemcpyAsync(vec_out + e.off, e.src, (size_t)e.n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "dstage_pack: vector copy failed");
  hipLaunchKernelGGL(stage_posw_pack_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, st, p->pos_w, reinterpret_cast<unsigned*>(vec_out + 17 * C), 9 * C);          // the 3 x 3 taps as bf16 pair words
  LMV_CHECK_LAUNCH("dstage_pack");
  return LMV_OK;
}

int lmv_dstage_fwd(const lmv_dstage_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !x || !c || !x_out || !c_out || !workspace) LMV_FAIL(LMV_ERR_SHAPE, "dstage_fwd: null argument");
  if (d->dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "dstage_fwd: bf16 only");
  const int nw = ds_variant(d->C, d->heads, d->hidden, d->H, d->W, d->M);
  if (!nw) LMV_FAIL(LMV_ERR_DTYPE, "dstage_fwd: unsupported stage shape");
  if (d->B <= 0 || d->nblocks <= 0 || !d->wpk || !d->vec) LMV_FAIL(LMV_ERR_SHAPE, "dstage_fwd: bad descriptor");
  if (d->kind < 0 || d->kind > 2) LMV_FAIL(LMV_ERR_SHAPE, "dstage_fwd: kind must be 0 (D blocks), 1 (C blocks) or 2 (S blocks)");
  if ((x == x_out && d->kind != 1) || c == c_out) LMV_FAIL(LMV_ERR_SHAPE, "dstage_fwd: outputs must not alias the inputs");
  const void* ptrs[] = {x, c, x_out, c_out, workspace, d->wpk, d->vec};
  for (const void* q : ptrs) if (!lmv_aligned16(q)) LMV_FAIL(LMV_ERR_SHAPE, "dstage_fwd: pointers must be 16-byte aligned");
  if (workspace_bytes < lmv_dstage_workspace_bytes(d->B, d->C)) LMV_FAIL(LMV_ERR_WORKSPACE, "dstage_fwd: workspace too small");
  // (timing probe.  "D" blocks: qkv1 (the k third folded away: 2 C^2), proj_x, the MLP per image token; the meta tokens' qkv2, q~, proj_c, MLP; both attention directions.  "C": k / v folded or computed (C^2), meta path)
  const double Cd = d->C, N = (double)d->H * d->W, M = d->M;
  const double fl_img = d->kind ? N * Cd * Cd * 2.0 : N * Cd * 11.0 * Cd * 2.0, fl_meta = M * Cd * (d->kind ? 11.0 : 13.0) * Cd * 2.0, fl_att = (Cd / 32.0) * N * M * 32.0 * (d->kind ? 4.0 : 8.0);
  LmvTimedLaunch timed(stream, (double)d->B * d->nblocks * (fl_img + fl_meta + fl_att), (double)d->B * (N * (d->kind ? 1.0 : 2.0) + 2.0 * M) * Cd * 2.0 + (double)d->nblocks * 18.0 * Cd * Cd * 2.0, 2);
#define DS_LAUNCH(NW, GW, CT) ds_launch<NW, GW, CT>(d, x, c, x_out, c_out, workspace, (hipStream_t)stream)
  return DS_DISPATCH(nw, DS_LAUNCH, LMV_ERR_DTYPE);
}
