// attn_internal.h -- shared between attn.hip (dispatch, fp32 / generic kernels) and attn_mfma.hip (bf16 MFMA kernels)
#pragma once
#include "common.h"

struct AttnArgs {
  const void* q; const void* k; const void* v; void* o; float* lse; const void* d_o;
  void* dq; void* dk; void* dv;
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
  int B, H, Lq, Lk;
  float scale;
};

bool lmv_attn_mfma_supported(const AttnArgs& a);          // bf16, Lk <= 224
size_t lmv_attn_mfma_bwd_acc_bytes(const AttnArgs& a);    // fp32 scratch for split query ranges
int lmv_attn_mfma_fwd(const AttnArgs& a, hipStream_t st);
int lmv_attn_mfma_bwd(const AttnArgs& a, float* delta, float* acc, hipStream_t st);   // delta: B*H*Lq floats of scratch (written by the dQ kernel)
// <= 16 queries over > 224 keys (bf16): split-key MFMA kernels
bool lmv_attn_mfma_fewq_supported(const AttnArgs& a);
int lmv_attn_mfma_fewq_nsplit(const AttnArgs& a);
int lmv_attn_mfma_fewq_fwd(const AttnArgs& a, float* part, hipStream_t st);
size_t lmv_attn_mfma_fewq_bwd_acc_bytes(const AttnArgs& a);   // fp32 dQ slabs, one per workgroup of the key split
int lmv_attn_mfma_fewq_bwd(const AttnArgs& a, float* acc, hipStream_t st);
// many queries over 225..640 keys (bf16): whole-row kernels with K / V of a (b, h) in dynamic LDS
bool lmv_attn_mfma_long_supported(const AttnArgs& a);
int lmv_attn_mfma_long_fwd(const AttnArgs& a, hipStream_t st);
int lmv_attn_mfma_long_bwd(const AttnArgs& a, float* delta, hipStream_t st);
// two independent problems (same B, H) in one launch: returns 1 when the merged kernel ran, 0 when the shapes do not qualify
int lmv_attn_mfma_fwd_pair(const AttnArgs& a1, const AttnArgs& a2, hipStream_t st);
int lmv_attn_mfma_bwd_pair(const AttnArgs& a1, const AttnArgs& a2, hipStream_t st);
