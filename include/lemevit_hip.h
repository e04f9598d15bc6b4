/* lemevit_hip.h -- C ABI of liblemevit_hip.so, the MI355X (gfx950) kernels behind the LeMeViT
 * backbone hot path.
 *
 * The reference (ViTAE-Transformer/LeMeViT) has no FFI: its hot path is PyTorch calls inside
 * models/lemevit.py, dispatched through the attention-backend seam at models/lemevit.py:32-52.
 * Each entry point below replaces the torch calls of the cited reference lines; the Python
 * binding a maintainer would add is shown in INTEGRATION.md (ctypes, as lemevit_amd/_lib.py).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator) and must stay alive until `stream` reaches the kernel;
 *   - tensors are contiguous row-major, token-major [B, L, C]; 16-byte aligned;
 *   - dtype: LMV_F32 = 0, LMV_BF16 = 1 (activations and matrix weights); vectors (bias, LayerNorm
 *     affine, statistics, DropPath scales) and gradient accumulators are always fp32;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *   - return 0 on success, else LMV_ERR_* (message via lmv_last_error()); nothing throws;
 *   - re-entrant: entry points may be called from several host threads and on any streams; the device is whatever the caller made current.
 *     Per-call state lives in the caller's buffers.  What the library keeps per PROCESS, and is therefore not "stateless", is listed here:
 *       * the tuning switches (lmv_config_set: process-wide, unsynchronised -- set them between launches);
 *       * one sticky error word per device (pinned host memory) that the persistent stage kernels raise (lmv_stage_error_count);
 *       * the measurement aids lmv_debug_launch_timing / lmv_stem_debug_timing (single host thread, never in production);
 *       * per-device caches of occupancy queries and of the hipFuncSetAttribute calls (write-once, atomics).
 */
#ifndef LEMEVIT_HIP_H
#define LEMEVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMV_ABI_VERSION 12

enum { LMV_F32 = 0, LMV_BF16 = 1 };
enum {
  LMV_OK = 0,
  LMV_ERR_SHAPE = -1,      /* bad shape / alignment */
  LMV_ERR_DTYPE = -2,      /* unsupported dtype / arch */
  LMV_ERR_WORKSPACE = -3,  /* workspace too small */
  LMV_ERR_LAUNCH = -4      /* HIP launch error (hipGetLastError) */
};

int lmv_abi_version(void);
const char* lmv_last_error(void);
/* Tuning switches (A/B runs, parity tests of alternative code paths): these two change / read a switch at run time.  Five of them are also read from the environment, ONCE, when
 * the library is loaded -- never on a launch path: LMV_GEMM_W8, LMV_GEMM_RS, LMV_GEMM_WN, LMV_DW_TARGET_BLOCKS, LMV_STAGE_TICKET_SKEW.  Keys: "gemm_bk", "gemm_bk32_tiles", "dw_bk",
 * "dw_target_blocks", "gemm_no_dma", "gemm_w8", "gemm_cumap", "gemm_nst", "gemm_nst_dw", "gemm_rs", "dwconv_v", "mlp_tm", "attn_pv16",
 * "attn_fuse_dq", "attn_fused_bwd", "attn_pair", "ln_bwd_blocks", "ln_bwd_minrows", "stage_ticket_skew" (test switch) (lemevit_amd/csrc/common.h: LmvConfig).
 * Process-wide, not synchronised: set them between launches. */
int lmv_config_set(const char* key, int value);
int lmv_config_get(const char* key, int* value);

/* ------------------------------------------------------------------------------------------
 * Linear layers (nn.Linear calls of models/lemevit.py:200,205,289,291,299,302,478-479,486 and
 * the MLP :526-530).  One launch runs up to two independent "problems" that share N and K (the
 * image-token matrix and the meta-token matrix of a block share LN/MLP -- and in S-blocks
 * attention -- weights, :560-564,:632-635).
 *
 *   fwd : out[r, n] = res[r, n] + row_scale[r / rows_per_sample] * act(sum_k a[r,k] w[n,k] + bias[n])
 *         (act == LMV_ACT_GELU_GRAD: out[r, n] = row_scale * (sum_k a[r,k] w[n,k] + bias[n]) * gelu'(aux[r, n]), no residual: a dX through
 *          a TRANSPOSED weight copy)
 *   dx  : out[r, k] = (sum_n a[r,n] w[n,k]) (* gelu'(aux[r,k]) if act == LMV_ACT_GELU_GRAD)
 *   dw  : dw[n, k] += sum_r dy[r,n] x[r,k] ;  db[n] += sum_r dy[r,n]      (fp32; split-K partial slabs in
 *         `workspace` + a reduce kernel: deterministic, no atomics)
 * ------------------------------------------------------------------------------------------ */
enum { LMV_ACT_NONE = 0, LMV_ACT_GELU = 1, LMV_ACT_GELU_GRAD = 2 };

typedef struct {
  const void* a;           /* fwd: x [rows, K]; dx: dy [rows, N]; dw: dy [rows, N]            */
  const void* w;           /* fwd/dx: weight [N, K];               dw: x  [rows, K]            */
  const float* bias;       /* fwd: [N] or NULL                                                  */
  const void* res;         /* fwd: residual [rows, N] or NULL (dx: [rows, K] added to out)      */
  const float* row_scale;  /* per-sample DropPath scale [rows / rows_per_sample] or NULL        */
  const void* aux;         /* dx with GELU_GRAD: pre-activation u [rows, K]                     */
  void* out;               /* fwd: [rows, N]; dx: [rows, K]; dw: fp32 dW [N, K] (accumulated)   */
  void* out_pre;           /* fwd: optional copy of the pre-activation (training) or NULL       */
  float* bias_grad;        /* dw: fp32 db [N] (accumulated) or NULL                             */
  int64_t rows;
  int32_t rows_per_sample; /* tokens per image for row_scale (ignored when row_scale == NULL)   */
  int32_t _pad;
} lmv_linear_problem;

int lmv_linear_fwd(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream);
int lmv_linear_dx(const lmv_linear_problem* p, int nproblems, int N, int K, int act, int dtype, void* stream);
size_t lmv_linear_dw_workspace_bytes(const lmv_linear_problem* p, int nproblems, int N, int K, int dtype);
int lmv_linear_dw(const lmv_linear_problem* p, int nproblems, int N, int K, void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* Deferred form: only the split-K GEMM is launched; the partial slabs stay in `workspace` (which must then stay untouched until
 * they are summed) and up to 2 reduce segments describing them are written to segs[0 .. *nsegs).  lmv_reduce_batch sums the slabs
 * of any number of segments -- the weight gradients of a whole block -- into their out_w / out_b (accumulating, fixed summation
 * order: bit-identical to lmv_linear_dw) in ONE launch per LMV_REDUCE_MAX_SEGS segments. */
#define LMV_REDUCE_MAX_SEGS 12
#define LMV_REDUCE_SLABS 0   /* split-K slabs of a weight-gradient GEMM (summation tree of lmv_linear_dw)                              */
#define LMV_REDUCE_ROWS  1   /* per-workgroup partial rows of a column reduction: LayerNorm dgamma | dbeta (lmv_layernorm_bwd_partial,  */
                             /* mode 0) or the depth-wise convolution's [10][C] tap sums (lmv_dwconv3x3_bwd_weight_partial, mode 1:     */
                             /* partial column tap * C + c -> out_w[c * 9 + tap], tap 9 -> out_b[c]); summation tree of the stand-alone */
                             /* reduce of those ops, so the merged launch is bit-identical to them                                       */
typedef struct {
  const float* ws;        /* first slab of the segment: [nslabs][slab_stride] fp32, a slab = [nw weights | nb bias sums]   */
  float* out_w;           /* fp32 [nw], accumulated                                                                          */
  float* out_b;           /* fp32 [nb], accumulated; NULL = no bias gradient                                                  */
  int64_t slab_stride;    /* floats between consecutive slabs (LMV_REDUCE_ROWS: row width = nw + nb, or 10 * nw in mode 1)    */
  int64_t nw;
  int32_t nslabs, nb;
  int32_t kind, mode;     /* LMV_REDUCE_SLABS (mode ignored) or LMV_REDUCE_ROWS with the output mapping `mode`                */
} lmv_reduce_seg;
int lmv_linear_dw_partial(const lmv_linear_problem* p, int nproblems, int N, int K, void* workspace, size_t workspace_bytes, int dtype, void* stream,
                          lmv_reduce_seg* segs, int* nsegs);
int lmv_reduce_batch(const lmv_reduce_seg* segs, int nsegs, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused block entry points (`ln_linear`, `mlp_fused`, `attn_out_proj_residual`).
 *
 * LayerNorm folded into the Linear that consumes it (norm1 -> qkv / q / kv, models/lemevit.py:560,599,632,634; norm2 -> mlp.0,
 * :563-564,:633,635).  With x_hat = (x - mean) rstd:
 *     LN(x) W^T + b  =  rstd (x W'^T - mean colsum(W'))  +  b',      W' = gamma . W  (rounded to `dtype`),  b' = b + W beta
 * so the GEMM reads the RAW residual-stream rows and the normalised copy never exists in HBM.
 *   lmv_ln_fold       : (W fp32 [N, K], bias [N] or NULL, gamma [K], beta [K]) -> wf [N, K] in `dtype`, colsum [N], bf [N] (fp32);
 *                       colsum is taken over the ROUNDED wf, so the mean term cancels exactly.  Run once per weight version.
 *   lmv_ln_linear_fwd : out[r, n] = act(LN(x[r, :]) . W[n, :] + b[n]) from the folded operands; the row statistics are accumulated
 *                       from the MFMA operand fragments inside the k-loop (one-pass, fp32).  p[i].a = x, p[i].w = wf, p[i].bias = bf,
 *                       p[i].aux = colsum (fp32 [N]); res / row_scale / out_pre as in lmv_linear_fwd.
 *   lmv_mlp_fused_fwd : out = x + row_scale * fc2(GELU(fc1(LN(x))))  -- the whole MLP half of a LeMeBlock in ONE kernel: a workgroup
 *                       keeps 128 token rows in LDS, streams the weights, and the 4C-wide hidden activations never leave the chip
 *                       (C in + C out per token instead of 17 C).  bf16, C in {64, 96, 128, 192, 256, 320, 384}, hidden % 128 == 0
 *                       (lmv_mlp_fused_supported); inference form: nothing is saved for a backward pass.
 *   lmv_attn_out_proj_residual : out = res + row_scale * (attn_out W^T + b) -- the output projection of an attention module with the
 *                       block's residual add and DropPath folded into the GEMM epilogue (:205 + :562 / :601 / :632,634).
 * ------------------------------------------------------------------------------------------ */
int lmv_ln_fold(const float* w, const float* bias, const float* gamma, const float* beta, int N, int K, void* wf, float* colsum, float* bf,
                int dtype, void* stream);
int lmv_ln_linear_fwd(const lmv_linear_problem* p, int nproblems, int N, int K, float eps, int act, int dtype, void* stream);
typedef struct {
  const void* x;            /* [rows, C]: input of the MLP half (= its residual)                 */
  void* out;                /* [rows, C]                                                          */
  const float* row_scale;   /* per-sample DropPath scale [rows / rows_per_sample] or NULL        */
  int64_t rows;
  int32_t rows_per_sample;
  int32_t _pad;
} lmv_mlp_problem;
typedef struct {
  const void* w1f; const float* colsum1; const float* b1f;   /* lmv_ln_fold(mlp.0, norm2): [hidden, C], [hidden], [hidden] */
  const void* w2; const float* b2;                           /* mlp.3: [C, hidden] in `dtype`, [C] fp32                    */
} lmv_mlp_weights;
int lmv_mlp_fused_supported(int C, int hidden, int dtype);
/* Test hook: y[i] = the activation lmv_mlp_fused_fwd applies to the hidden values (a degree-7 minimax fit of GELU in packed fp32,
 * |y - GELU_erf(x)| <= 1.9e-4 absolute, exact 0 / identity beyond |x| >= 4), for x, y fp32 [n] on the device. */
int lmv_gelu_poly_eval(const float* x, float* y, int64_t n, void* stream);
int lmv_mlp_fused_fwd(const lmv_mlp_problem* p, int nproblems, const lmv_mlp_weights* w, int C, int hidden, float eps, int dtype, void* stream);
int lmv_attn_out_proj_residual(const lmv_linear_problem* p, int nproblems, int C, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (nn.LayerNorm, models/lemevit.py:513,525 eps 1e-6; :731-743,774
 * eps 1e-5).  One launch normalises up to TWO row segments with the same (gamma, beta): the image-token
 * and the meta-token matrix of a block share norm1 / norm2 (:560-564,:632-635).
 *   fwd: y = LN(x); stats = [rows, 2] fp32 (mean, rstd), written when non-NULL.
 *   bwd: dx = dres + LN'(dy)   (dres may be NULL);  dgamma/dbeta are ACCUMULATED (fp32; per-workgroup partials
 *        in `workspace` + a reduce kernel, no atomics).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* x;      /* [rows, C] input of the forward pass                               */
  void* y;            /* fwd: output [rows, C]                                              */
  float* stats;       /* [rows, 2] (mean, rstd): written by fwd (may be NULL), read by bwd  */
  const void* dy;     /* bwd: gradient of y                                                 */
  const void* dres;   /* bwd: optional residual-path gradient added to dx (or NULL)         */
  void* dx;           /* bwd: output [rows, C]                                              */
  int64_t rows;
  /* bwd, optional (all three or none): a second output dx_scaled[r, :] = dx[r, :] * dx_scale[r / rows_per_sample] -- the
   * DropPath-scaled gradient the NEXT backward stage consumes (saves a separate lmv_row_scale pass over dx) */
  const float* dx_scale;
  void* dx_scaled;
  int64_t rows_per_sample;
} lmv_ln_segment;

int lmv_layernorm_fwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, int C, float eps,
                      int dtype, void* stream);
/* y = GELU(LayerNorm(x)) in one pass and its backward (dy is multiplied by GELU'(LayerNorm(x)), recomputed from x, gamma, beta):
 * the Linear -> LayerNorm -> GELU -> Linear -> LayerNorm meta-token MLP of every stage (models/lemevit.py:731-743). */
int lmv_layernorm_gelu_fwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, int C, float eps,
                           int dtype, void* stream);
int lmv_layernorm_gelu_bwd(const lmv_ln_segment* seg, int nseg, const float* gamma, const float* beta, float* dgamma, float* dbeta, int C,
                           void* workspace, size_t workspace_bytes, int dtype, void* stream);
size_t lmv_layernorm_bwd_workspace_bytes(int64_t total_rows, int C, int dtype);
int lmv_layernorm_bwd(const lmv_ln_segment* seg, int nseg, const float* gamma, float* dgamma, float* dbeta, int C,
                      void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* The same in two calls: _partial writes dx and leaves `*partial_rows` per-workgroup rows of (dgamma | dbeta) partial sums in
 * `workspace`; _reduce accumulates them into dgamma / dbeta -- on any stream that is ordered behind _partial (the block scheduler
 * keeps it off the critical path).  Results are bit-identical to lmv_layernorm_bwd. */
int lmv_layernorm_bwd_partial(const lmv_ln_segment* seg, int nseg, const float* gamma, int C, void* workspace, size_t workspace_bytes,
                              int* partial_rows, int dtype, void* stream);
int lmv_layernorm_bwd_reduce(const void* workspace, int partial_rows, int C, float* dgamma, float* dbeta, void* stream);
/* The dX of a Linear fused with the LayerNorm backward of the Linear's INPUT (models/lemevit.py:560,563: norm -> Linear; csrc/wngemm.hip,
 * bf16, C = 384, N % 64 == 0: lmv_linear_dx_ln_bwd_supported):
 *     dy = dY wt^T   (wt = the Linear's weight TRANSPOSED, [C, N]: lmv_transpose_batch),     dx = dres + LN'(dy),     dx_scaled = dx * dx_scale[sample]
 * p[i].a = dY [rows, N], p[i].w = wt; seg[i] as for lmv_layernorm_bwd_partial (x, stats, dres, dx, rows, dx_scale / dx_scaled /
 * rows_per_sample; seg[i].dy is ignored -- dy never reaches memory, the LayerNorm sees it in fp32).  `*partial_rows` rows of (dgamma | dbeta)
 * partial sums are left in `workspace` for lmv_layernorm_bwd_reduce / an LMV_REDUCE_ROWS segment of lmv_reduce_batch. */
/* The forward mirror: a Linear with the residual epilogue followed by the LayerNorm of its output (proj + norm2, models/lemevit.py:562-563,
 * 632-635) in one launch:  out = res + row_scale (a W^T + bias);  seg[i].y = LN(out) (of the ROUNDED out, as a separate launch would read
 * it), seg[i].stats = (mean, rstd) when non-NULL.  bf16, N = 384, K % 64 == 0 (lmv_linear_res_ln_fwd_supported). */
/* LayerNorm -> Linear with the EXACT LayerNorm in front (no folding), for the C = 96 layers (norm1 -> qkv1 / qkv2 / kv / q,
 * models/lemevit.py:560,599; csrc/rswgemm.hip: the whole weight matrix resident in LDS, every wave streams 32-row panels through registers):
 *     seg[i].y = LN(p[i].a)  (written when non-NULL: training keeps it for the weight gradient),  seg[i].stats = (mean, rstd) (when non-NULL),
 *     p[i].out = seg[i].y W^T + bias.      gamma == NULL: plain Linear.   bf16, K = 96, N % 32 == 0, N <= 384 (lmv_ln_linear_exact_fwd_supported). */
int lmv_ln_linear_exact_fwd_supported(int N, int K, int dtype);
int lmv_ln_linear_exact_fwd(const lmv_linear_problem* p, const lmv_ln_segment* seg, int nproblems, int N, int K, const float* gamma, const float* beta,
                            float eps, int dtype, void* stream);
int lmv_linear_res_ln_fwd_supported(int N, int K, int dtype);
int lmv_linear_res_ln_fwd(const lmv_linear_problem* p, const lmv_ln_segment* seg, int nproblems, int N, int K, const float* gamma, const float* beta,
                          float eps, int dtype, void* stream);
int lmv_linear_dx_ln_bwd_supported(int C, int N, int dtype);
size_t lmv_linear_dx_ln_bwd_workspace_bytes(int64_t total_rows, int C);
int lmv_linear_dx_ln_bwd(const lmv_linear_problem* p, const lmv_ln_segment* seg, int nproblems, int C, int N, const float* gamma,
                         void* workspace, size_t workspace_bytes, int* partial_rows, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode BatchNorm2d (+ exact GELU) over a channels-last feature map viewed as [rows = B*H*W][C]: the BatchNorm2d
 * (-> GELU) of the stem and of every stage transition (models/lemevit.py:698-704, 714-717) and the final `norm` (:773, 822).
 * Replaces torch.nn.BatchNorm2d.forward in training mode (+ the nn.GELU after the first stem BatchNorm) and their autograd.
 *   fwd: batch mean / biased variance per channel (fp32 partial sums, fp64 combine); y = (x - mean) * rstd * gamma + beta
 *        (act = LMV_ACT_GELU: GELU of that); stats[0:C] = mean, stats[C:2C] = rstd (saved for the backward);
 *        running_mean / running_var (may be NULL) updated with `momentum` and the UNBIASED variance, as torch.
 *   bwd: dgamma / dbeta are WRITTEN (not accumulated); dx = gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' * xhat)),
 *        dy' = dy * GELU'(xhat * gamma + beta) when act = LMV_ACT_GELU.
 * C % 8 == 0 (bf16) / C % 4 == 0 (fp32), C <= 2048 / 1024; workspace: lmv_batchnorm_workspace_bytes(C).
 * ------------------------------------------------------------------------------------------ */
size_t lmv_batchnorm_workspace_bytes(int C);
int lmv_batchnorm_train_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                            float eps, int act, void* y, float* stats, int64_t rows, int C, void* workspace, size_t workspace_bytes,
                            int dtype, void* stream);
int lmv_batchnorm_train_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats, int act, void* dx,
                            float* dgamma, float* dbeta, int64_t rows, int C, void* workspace, size_t workspace_bytes, int dtype,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Conditional position embedding: y = x + dwconv3x3(x) + bias, NHWC (models/lemevit.py:510,546).
 * weight is the reference's [C, 1, 3, 3] fp32 tensor.
 *   bwd_data  : dx = dy + dwconv3x3^T(dy)
 *   bwd_weight: dw[C,1,3,3] += ..., db[C] += ...     (fp32; per-workgroup partials in `workspace` + reduce)
 * ------------------------------------------------------------------------------------------ */
int lmv_dwconv3x3_residual_fwd(const void* x, const float* weight, const float* bias, void* y,
                               int B, int H, int W, int C, int dtype, void* stream);
int lmv_dwconv3x3_residual_bwd_data(const void* dy, const float* weight, void* dx,
                                    int B, int H, int W, int C, int dtype, void* stream);
size_t lmv_dwconv3x3_bwd_weight_workspace_bytes(int B, int H, int W, int C, int dtype);
int lmv_dwconv3x3_bwd_weight(const void* dy, const void* x, float* dweight, float* dbias,
                             int B, int H, int W, int C, void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* Deferred form: only the tap-sum kernel runs; `*partial_rows` rows of [10][C] partial sums stay in `workspace` and are accumulated into
 * dweight / dbias later by lmv_reduce_batch (segment kind LMV_REDUCE_ROWS, mode 1, nw = C, nb = C, slab_stride = 10 C, nslabs = rows). */
int lmv_dwconv3x3_bwd_weight_partial(const void* dy, const void* x, int B, int H, int W, int C, void* workspace, size_t workspace_bytes,
                                     int* partial_rows, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention cores, head dim 32 (all registered variants, models/lemevit.py:851,881,911).
 * q/k/v/o are addressed as  ptr + b * batch_stride + l * row_stride + head * 32  (strides in
 * ELEMENTS), so the packed projections of the reference (qkv [B,L,3C] :200-202, kv [B,N,2C]
 * :479-482) are consumed in place -- no 'x B h N d' re-pack (:201,290,292,481).
 *   o[b,l,h,:] = softmax_j(scale * q[b,l,h,:] . k[b,j,h,:]) v[b,j,h,:]       (:54-63)
 *   lse[b,h,l] = log sum_j exp(scale * q.k)   (fp32, kept for the backward pass)
 * Lq <= 16 (meta-token queries over image-token keys, :300,:484) takes the split-key path and
 * needs `workspace` (lmv_attn_workspace_bytes).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* q; const void* k; const void* v;  /* inputs                                       */
  void* o;                                      /* fwd: output; bwd: the saved forward output    */
  float* lse;                                   /* [B, H, Lq] fp32                               */
  const void* d_o;                              /* bwd: grad of o (strides of o)                 */
  void* dq; void* dk; void* dv;                 /* bwd: grads (strides of q / k / v)             */
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
  int32_t B, H, Lq, Lk;
  float scale;
  int32_t _pad;
} lmv_attn_desc;

size_t lmv_attn_workspace_bytes(int B, int H, int Lq, int Lk, int backward);
int lmv_attn_fwd(const lmv_attn_desc* d, void* workspace, size_t workspace_bytes, int dtype, void* stream);
int lmv_attn_bwd(const lmv_attn_desc* d, void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* Two independent problems d[0], d[1] with the same B and H -- the image-token and the meta-token self-attention of an "S" block
 * (models/lemevit.py:632,634) -- in ONE launch where the shapes allow (bf16, 196 / 49 + 16 tokens), otherwise as two launches.
 * The workspace must be large enough for either problem (max of lmv_attn_workspace_bytes). */
int lmv_attn_fwd_pair(const lmv_attn_desc* d, void* workspace, size_t workspace_bytes, int dtype, void* stream);
int lmv_attn_bwd_pair(const lmv_attn_desc* d, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* Named cores of the reference seam; thin wrappers that fill an lmv_attn_desc.
 *   sa : StandardAttention      qkv [B,L,3C] -> o [B,L,C]                        (:199-205)
 *   ca : CrossAttention         q [B,M,C], kv [B,N,2C] -> o [B,M,C]              (:477-486)
 *   dca: DualCrossAttention     qkv1 [B,N,3C], qkv2 [B,M,3C] -> ox [B,N,C], oc [B,M,C]
 *        with scale_x = log_N(M) C^-1/2, scale_c = C^-1/2                        (:235,255-256,288-302) */
int lmv_sa_core_fwd(const void* qkv, void* o, float* lse, int B, int L, int C, void* ws, size_t ws_bytes, int dtype, void* stream);
int lmv_ca_core_fwd(const void* q, const void* kv, void* o, float* lse, int B, int M, int N, int C,
                    void* ws, size_t ws_bytes, int dtype, void* stream);
int lmv_dca_core_fwd(const void* qkv1, const void* qkv2, void* ox, void* oc, float* lse_x, float* lse_c,
                     int B, int N, int M, int C, void* ws, size_t ws_bytes, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small utilities used by the host side
 * ------------------------------------------------------------------------------------------ */
/* dst[i] = (dtype_dst) src[i]   (fp32 master weights -> bf16 compute copies and back) */
int lmv_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* y[r, :] = x[r, :] * scale[r / rows_per_sample]   (DropPath on a gradient) */
/* Stem, first convolution (models/lemevit.py:713, Conv2d(3, C/2, kernel 3, stride 2, padding 1)) lowered to a GEMM:
 * patches[(b, ho, wo)][ci * 9 + ky * 3 + kx] = x[b][ci][2 ho - 1 + ky][2 wo - 1 + kx] (zero outside the image), columns
 * 27..31 zero -> a [B * ceil(H/2) * ceil(W/2), 32] matrix in `dtype`.  x is read through its element strides
 * (sb, sc, sh, sw), so NCHW and channels-last batches of fp32 or bf16 images are accepted as they are.
 * lmv_linear_fwd(patches, W[Cout, 32]) is then the convolution (NHWC output), lmv_linear_dw its weight gradient. */
int lmv_im2col3x3s2_c3(const void* x, int x_dtype, void* patches, int dtype, int B, int H, int W, int64_t sb, int64_t sc, int64_t sh,
                       int64_t sw, void* stream);
int lmv_row_scale(const void* x, const float* scale, void* y, int64_t rows, int C, int rows_per_sample, int dtype, void* stream);
/* The same for up to two row segments in ONE launch (the x and c gradients of a block, each with its own DropPath vector). */
typedef struct lmv_row_scale_segment {
  const void* x; const float* scale; void* y;
  int64_t rows; int rows_per_sample;
} lmv_row_scale_segment;
int lmv_row_scale_multi(const lmv_row_scale_segment* seg, int nseg, int C, int dtype, void* stream);
/* dst[c][r] = src[r][c] for any number of bf16 matrices (src [rows, cols] -> dst [cols, rows]) in one launch per LMV_TRANSPOSE_MAX_SEGS:
 * the transposed weight copies that let a dX run as a forward-form GEMM (lmv_block_desc.fc2_wt). */
#define LMV_TRANSPOSE_MAX_SEGS 48
typedef struct { const void* src; void* dst; int32_t rows, cols; } lmv_transpose_seg;
int lmv_transpose_batch(const lmv_transpose_seg* segs, int nsegs, int dtype, void* stream);
/* Dense 3x3 / stride-2 / padding-1 convolutions on channels-last maps -- the second stem convolution and the stage transitions
 * (models/lemevit.py:701-703, :714-717) -- lowered to the block GEMM:
 *   patches[(b, ho, wo)][(ky * 3 + kx) * C + ci] = x[b, 2 ho - 1 + ky, 2 wo - 1 + kx, ci]  (zero outside the map and in the padding
 *   columns 9 C .. KP - 1), x = [B, H, W, C] NHWC, patches = [B * ceil(H/2) * ceil(W/2), KP], C % 8 == 0, KP >= 9 C, KP % 8 == 0.
 * lmv_linear_fwd(patches, Wm[Cout, KP]) is the convolution (NHWC output), lmv_linear_dw its weight gradient, and
 * lmv_col2im3x3s2_nhwc(lmv_linear_dx(dY, Wm)) its data gradient (a gather over the <= 4 output pixels that read an input pixel). */
int lmv_im2col3x3s2_nhwc(const void* x, void* patches, int B, int H, int W, int C, int KP, int dtype, void* stream);
int lmv_col2im3x3s2_nhwc(const void* dpatches, void* dx, int B, int H, int W, int C, int KP, int dtype, void* stream);
/* The same convolution WITHOUT the patch matrix (round 6; bf16): an implicit GEMM -- the LDS-DMA loads of the GEMM kernels gather the patch elements straight from the
 * NHWC map x [B, H, W, Cin] (zeros for the padding taps and for the columns behind 9 Cin), so the forward pass and the weight gradient read the map instead of writing and
 * re-reading a [B Ho Wo, KP] matrix 2.25 x its size.  wm / dwm: [Cout, KP] in lmv_im2col3x3s2_nhwc's column order ((ky * 3 + kx) * Cin + ci), KP >= 9 Cin a multiple of 64;
 * B Ho Wo must be a multiple of 64 (whole k-tiles of the weight gradient); y [B Ho Wo, Cout] = patches wm^T + bias (act: LMV_ACT_NONE / LMV_ACT_GELU);
 * dwm += dy^T patches, dbias += column sums of dy (fp32, accumulated; split-K slabs in `workspace`, fixed summation order as lmv_linear_dw).  The data gradient stays
 * lmv_linear_dx + lmv_col2im3x3s2_nhwc.  Replaces models/lemevit.py:701-703, :714-717 (nn.Conv2d(.., 3, 2, 1)). */
int lmv_conv3x3s2_fwd(const void* x, const void* wm, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int KP, int act, int dtype, void* stream);
size_t lmv_conv3x3s2_dw_workspace_bytes(int B, int H, int W, int Cin, int Cout, int KP, int dtype);
int lmv_conv3x3s2_dw(const void* dy, const void* x, float* dwm, float* dbias, int B, int H, int W, int Cin, int Cout, int KP, void* workspace, size_t workspace_bytes, int dtype,
                     void* stream);
/* Classifier tail (models/lemevit.py:815-835, `x.flatten(2).mean(-1) + c.mean(1)`): out[b, :] = mean_l x[b, l, :] + mean_m c[b, m, :]
 * for token-major x [B, L, C] and c [B, M, C] (c may be NULL), out [B, C] in `dtype`; and its backward, the broadcast
 * dx[b, l, :] = g[b, :] / L, dc[b, m, :] = g[b, :] / M (dc may be NULL). */
int lmv_token_mean2_fwd(const void* x, int L, const void* c, int M, int C, int B, void* out, int dtype, void* stream);
int lmv_token_mean2_bwd(const void* g, void* dx, int L, void* dc, int M, int C, int B, int dtype, void* stream);
/* Inference tail (models/lemevit.py:815, 825 with the final BatchNorm in eval mode): BatchNorm with running statistics is affine per channel and
 * commutes with the spatial mean, so out[b, :] = xscale * mean_l x[b, l, :] + xshift + mean_m c[b, m, :] with
 * xscale = gamma / sqrt(running_var + eps), xshift = beta - running_mean * xscale (fp32 [C]). */
int lmv_token_mean2_affine_fwd(const void* x, int L, const void* c, int M, int C, int B, const float* xscale, const float* xshift, void* out, int dtype,
                               void* stream);
/* Fused multi-tensor AdamW over a flat fp32 parameter / gradient / moment buffer
 * (decoupled weight decay, bias correction as torch.optim.AdamW; benchmark.py:559-561,587).
 * wd_mask (nullable): per-element 0/1 factor on weight_decay.  shadow_bf16 (nullable): bf16 copy of the updated parameters,
 * written in the same pass (the operand copy the block kernels read; no separate cast launches).  step_dev (nullable):
 * device int holding the 1-based step count -- read by the kernel instead of `step`, so a captured hipGraph replays with
 * the right bias correction. */
int lmv_adamw_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* wd_mask, void* shadow_bf16,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, const int* step_dev,
                   void* stream);
/* Exponential moving average of the weights over the same flat buffer (timm.utils.ModelEmaV2: main.py:316, engine.py model_ema.update(model)):
 * ema[i] = decay * ema[i] + (1 - decay) * param[i], n % 4 == 0, one launch (lemevit_amd.optim.ModelEma). */
int lmv_ema_flat(float* ema, const float* param, int64_t n, float decay, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-block schedules: ONE call enqueues every launch of a LeMeBlock (models/lemevit.py:500-660) on token-major tensors
 * x [B, H*W, C], c [B, M, C] -- `LeMeBlock.forward_with_x` ("S", :615-650), `forward_with_xc` ("D", :542-582), `forward_with_c`
 * ("C", :584-613; x is returned untouched by the caller, x_out / dx_out may be NULL).  Replaces the ~12 / ~30 per-op calls a Python
 * scheduler makes per block and pass (lemevit_amd/blocks.py keeps that schedule for the variants not covered here).
 *   params: matrices (attn_w, fc1_w, fc2_w) in `dtype`, vectors (biases, LayerNorm affine, pos_embed weight [C, 9] and bias) fp32.
 *     attn_w / attn_b by kind:  S: {qkv, proj}   D: {qkv1, qkv2, proj_x, proj_c}   C: {q, kv, proj}
 *   masks: per-sample DropPath scale vectors [B] fp32 or NULL, in the reference's draw order (S / D: x-attn, x-mlp, c-attn, c-mlp; C: c-attn, c-mlp).
 *   g_*: fp32 gradient accumulators of the same shapes (backward only; accumulated, split reductions: deterministic).
 * Memory: the library allocates nothing.  `arena` (lmv_block_arena_bytes) receives every forward intermediate and IS the saved state
 * of the backward pass when save != 0 (keep it until lmv_block_bwd; with save == 0 it is scratch).  `scratch`
 * (lmv_block_bwd_scratch_bytes) holds the backward temporaries; one buffer may serve all blocks.  side_stream (nullable): the
 * weight-gradient launches are enqueued there behind an event on `stream`, and `stream` waits for them before the call returns.
 * Inference (save == 0) with flags & LMV_BLOCK_FUSED takes the fused entry points above (LayerNorm folded into qkv / q / kv, the MLP
 * half in one kernel); the training form keeps the per-layer launches, whose intermediates the backward pass reads.
 * ------------------------------------------------------------------------------------------ */
enum { LMV_BLOCK_S = 0, LMV_BLOCK_D = 1, LMV_BLOCK_C = 2 };
enum { LMV_BLOCK_NO_JOIN = 1, LMV_BLOCK_FUSED = 2 };
typedef struct lmv_block_desc {
  int32_t kind, dtype, B, H, W, M, C, hidden;      /* hidden: MLP width (0 = 4 C) */
  float eps;                                       /* LayerNorm eps of norm1 / norm2 (1e-6, models/lemevit.py:513,525) */
  int32_t flags;                                   /* LMV_BLOCK_NO_JOIN: lmv_block_bwd leaves the side stream un-joined (the caller records an
                                                      event on it and makes `stream` wait later; scratch and arena must then outlive that wait) */
  const float* pos_w; const float* pos_b; const float* n1_w; const float* n1_b;
  const void* attn_w[4]; const float* attn_b[4];
  const float* n2_w; const float* n2_b; const void* fc1_w; const float* fc1_b; const void* fc2_w; const float* fc2_b;
  const float* masks[4];
  float* g_pos_w; float* g_pos_b; float* g_n1_w; float* g_n1_b;
  float* g_attn_w[4]; float* g_attn_b[4];
  float* g_n2_w; float* g_n2_b; float* g_fc1_w; float* g_fc1_b; float* g_fc2_w; float* g_fc2_b;
  /* LMV_BLOCK_FUSED (lmv_block_fwd with save == 0, bf16): lmv_ln_fold operands (folded weight, colsum, folded bias) of the Linears that
   * consume norm1 -- S: {qkv, -}, D: {qkv1, qkv2}, C: {q, kv} -- and of mlp.0 (norm2).  The block then runs lmv_ln_linear_fwd instead of
   * LayerNorm + Linear where that is the faster form and lmv_mlp_fused_fwd for the MLP half where lmv_mlp_fused_supported. */
  const void* fold_attn_w[2]; const float* fold_attn_s[2]; const float* fold_attn_b[2];
  const void* fold_fc1_w; const float* fold_fc1_s; const float* fold_fc1_b;
  /* optional (NULL = absent): mlp.3.weight TRANSPOSED, [hidden, C] in `dtype` (lmv_transpose_batch of fc2_w).  With it lmv_block_bwd runs the dX
   * of fc2 -- du = (dOut W2) * GELU'(u) -- as the forward-form GEMM  dOut [rows, C] x fc2_wt^T  with the GELU' epilogue, which the
   * register-stationary kernel (csrc/rsgemm.hip) takes for C = 192 / 384.  Must hold the same values as fc2_w. */
  const void* fc2_wt;
  /* optional (NULL = absent), S blocks: mlp.0.weight TRANSPOSED [C, hidden] and the attention weights attn_w[0] (qkv) / attn_w[1] (proj)
   * TRANSPOSED [C, 3C] / [C, C] in `dtype`.  With them lmv_block_bwd runs the dX of fc1 / qkv / proj as forward-form GEMMs
   * dY [rows, N] x wt^T -> [rows, C], which the whole-width kernel (csrc/wngemm.hip) takes for C = 384.  Same values as the weights. */
  const void* fc1_wt; const void* attn_wt[2];
} lmv_block_desc;
size_t lmv_block_arena_bytes(const lmv_block_desc* d);
size_t lmv_block_bwd_scratch_bytes(const lmv_block_desc* d);
int lmv_block_fwd(const lmv_block_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* arena, size_t arena_bytes, int save, void* stream);
/* The images [image0, image0 + nimages) of the same call: `d`, the tensors and the arena are those of the WHOLE batch (every tensor is
 * [rows, width] with the images outermost, so a range of images is a contiguous slice of each; the attention workspace of the arena is
 * sliced the same way).  The images of a batch do not interact inside a LeMeBlock (models/lemevit.py:500-650: LayerNorm, per-image
 * attention, per-row Linears), so calls over disjoint ranges that cover the batch, on ANY streams and in any order, leave the outputs and
 * the arena in the state one lmv_block_fwd call leaves them in (up to the kernel selection, which follows the row count) -- the training
 * forward runs them on concurrent streams so that the ramp and the tail of each launch are covered by another range's kernels;
 * lmv_block_bwd then consumes the arena as a whole. */
int lmv_block_fwd_range(const lmv_block_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* arena, size_t arena_bytes, int save,
                        int image0, int nimages, void* stream);
/* x, c: the block inputs of the forward call; dx_out / dc_out: gradients of x_out / c_out; dx / dc: gradients of x / c (written). */
int lmv_block_bwd(const lmv_block_desc* d, const void* x, const void* c, const void* arena, size_t arena_bytes, const void* dx_out, const void* dc_out,
                  void* dx, void* dc, void* scratch, size_t scratch_bytes, void* stream, void* side_stream);

/* ------------------------------------------------------------------------------------------
 * A whole run of "S" blocks as ONE persistent launch (csrc/sstage.hip; inference form, bf16): stage 3 of LeMeViT-Base -- 18 x
 * LeMeBlock.forward_with_x (models/lemevit.py:615-650, live :631-635; StandardAttention :185-205; MLP :526-530) on x [B, 196, 384] and
 * c [B, 16, 384].  The token rows stay on chip for the whole stage (two workgroups per image, residual stream in registers); only the
 * block weights stream from L2.  Replaces nblocks lmv_block_fwd(kind = S, save = 0) calls; same math, the residual stream is kept in
 * fp32 between the blocks instead of being rounded to bf16 after every residual add.
 *   lmv_sstage_supported: 1 where the kernel applies: C = 384 / 12 heads / hidden 1536 (LeMeViT-Base, Small-v2: 8 waves per workgroup) or C = 192 / 6 heads /
 *     hidden 768 (LeMeViT-Tiny: 4 waves, two workgroups per CU), 14 x 14 image tokens, 16 meta tokens, bf16.
 *   lmv_sstage_pack: one block's parameters (matrices bf16, vectors fp32, the reference's layouts: attn.qkv [3C, C], attn.proj [C, C],
 *     mlp.0 [4C, C], mlp.3 [C, 4C], pos_embed.weight [C, 9]) -> `wpk_out` (lmv_sstage_wpk_bytes: the matrices in MFMA-fragment order) and
 *     `vec_out` (lmv_sstage_vec_floats fp32).  The packed blocks of a stage are consecutive: block j at wpk + j * wpk_bytes, vec + j * vec_floats.
 *   lmv_sstage_fwd: x_out / c_out may alias x / c.  `workspace` (lmv_sstage_workspace_bytes(min(B, lmv_sstage_max_images(C)), C)) holds the K / V fragments and the
 *     grid rows the two halves of an image exchange, the parked residual registers, and the flags (reset by the call on `stream`).  The launch needs 2 * min(B, max images)
 *     co-resident workgroups (512 threads / 147 KB LDS, one per CU; C = 192: 256 threads / 74 KB, two per CU): concurrent calls on different streams are safe up to lmv_sstage_max_concurrent() at a time (each with its own workspace).
 * ------------------------------------------------------------------------------------------ */
typedef struct lmv_sstage_block_params {
  int32_t C, heads, hidden, _pad;
  const void* qkv_w; const void* proj_w; const void* fc1_w; const void* fc2_w;
  const float* n1_w; const float* n1_b; const float* qkv_b; const float* proj_b;
  const float* n2_w; const float* n2_b; const float* fc1_b; const float* fc2_b;
  const float* pos_w; const float* pos_b;
} lmv_sstage_block_params;
typedef struct lmv_sstage_desc {
  int32_t B, H, W, M, C, heads, hidden, nblocks, dtype;
  float eps;                                       /* LayerNorm eps of norm1 / norm2 (1e-6) */
  const void* wpk; const float* vec;               /* nblocks packed blocks (lmv_sstage_pack) */
  void* timing; int32_t timing_block, kind;        /* optional (NULL): uint64 s_memtime stamps [workgroup][8 waves][24] of block `timing_block` (tools/sstage_timeline.py); kind: 0 (lmv_dstage_fwd: 0 = "D" blocks, 1 = "C" blocks) */
} lmv_sstage_desc;
int lmv_sstage_supported(int C, int heads, int hidden, int H, int W, int M, int dtype);
size_t lmv_sstage_wpk_bytes(int C, int hidden);
size_t lmv_sstage_vec_floats(int C, int hidden);
size_t lmv_sstage_workspace_bytes(int B, int C);
int lmv_sstage_max_images(int C);                 /* images one launch takes: (workgroups of the instance the device holds) / 2, whole groups of 8 -- 128 at C = 384, 256 at C = 192 on an MI355X; size the workspace for min(B, this) */
int lmv_sstage_max_concurrent(int C);             /* lmv_sstage_fwd calls that may be in flight on different streams of a device at once (see lmv_dstage_max_concurrent); 0: none */
int lmv_sstage_pack(const lmv_sstage_block_params* p, void* wpk_out, float* vec_out, void* stream);
int lmv_sstage_fwd(const lmv_sstage_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * A whole run of "D" blocks as ONE persistent launch (csrc/dstage.hip; inference form, bf16): stages 1 and 2 of LeMeViT-Base -- 4 x
 * LeMeBlock.forward_with_xc (models/lemevit.py:542-582, live :560-564; DualCrossAttention :220-324, live :288-302; MLP :526-530) on
 * x [B, 784, 192] and c [B, 16, 192].  An image is 7 workgroups of 112 image tokens (4 grid rows) plus one workgroup for the 16 meta tokens,
 * all resident together; per block they exchange, through L2, the grid rows next to a cut, the meta tokens' K2 / V2 / (q2 W_k1) operand
 * fragments, and the per-workgroup softmax partials of the meta queries over the image keys.  Replaces nblocks lmv_block_fwd(kind = D, save = 0)
 * calls; same math (the image tokens' k projection is folded into the meta queries; its bias cancels in the softmax), residual stream fp32
 * between the blocks.
 *   lmv_dstage_supported: 1 for C = 192 (LeMeViT-Base / -Small stage 2) or 128 (LeMeViT-Tiny stage 2) at 28 x 28 image tokens (7 + 1 workgroups of 4 waves per image) and for
 *     C = 96 (Base / Small stage 1) or 64 (Tiny stage 1) at 56 x 56 (28 + 1 workgroups of 2 waves); heads = C / 32, hidden = 4 C, 16 meta tokens, bf16.
 *   lmv_dstage_pack: one block's parameters (matrices bf16, vectors fp32, the reference's layouts: attn.qkv1 / attn.qkv2 [3C, C], attn.proj_x /
 *     attn.proj_c [C, C], mlp.0 [4C, C], mlp.3 [C, 4C], pos_embed.weight [C, 9]) -> wpk_out / vec_out; blocks of a stage consecutive as for lmv_sstage_pack.
 *   lmv_dstage_fwd: x_out / c_out must NOT alias x / c.  `workspace`: lmv_dstage_workspace_bytes(B, C) (exchange buffers and flags of the image
 *     slots; flags reset by the call on `stream`; one workspace per concurrent call).  All 8 (29) workgroups of an image slot must be co-resident (256 threads / 74 KB LDS each, two per CU;
 *     128 / 37 KB, four per CU).  Concurrent calls on different streams are safe up to lmv_dstage_max_concurrent() at a time (lemevit_amd.graph.split_forward issues up to 4): slots are assigned by ticket
 *     (see "Residency" below), so every call holds at most 8 incomplete slots and the rest of the chip always runs complete ones, whatever the dispatch order.
 * ------------------------------------------------------------------------------------------ */
typedef struct lmv_dstage_block_params {
  int32_t C, heads, hidden, _pad;
  const void* qkv1_w; const void* qkv2_w; const void* projx_w; const void* projc_w; const void* fc1_w; const void* fc2_w;
  const float* n1_w; const float* n1_b; const float* qkv1_b; const float* qkv2_b; const float* projx_b; const float* projc_b;
  const float* n2_w; const float* n2_b; const float* fc1_b; const float* fc2_b;
  const float* pos_w; const float* pos_b;
} lmv_dstage_block_params;
/* kind = 2: a run of "S" blocks on a long sequence (LeMeBlock.forward_with_x, models/lemevit.py:615-650; C = 384, 24 x 24 image tokens: stage 3 of LeMeViT-Base at 384 x 384), packed as a D
 * block with qkv1 = qkv2 = attn.qkv and proj_x = proj_c = attn.proj (lemevit_amd/ops.py::s2stage_pack): 6 image-row workgroups + 1 meta workgroup per image, the image's keys and values cross
 * its workgroups through L2. */
typedef lmv_sstage_desc lmv_dstage_desc;          /* same fields; timing: uint64 stamps [workgroup][waves][16]; kind = 1: a run of "C" blocks (stage 0: LeMeBlock.forward_with_c,
                                                   * models/lemevit.py:584-612 -- only the meta tokens change: c += proj(softmax(q(n1 c) k(n1 x')^T / sqrt(32)) v(n1 x')), c += mlp(n2 c) with x' = x + dwconv(x);
                                                   * x passes through, x_out is not written and may be NULL-equivalent = x).  Packed as a D block with qkv1 = [0 | attn.kv], qkv2 = [attn.q | 0 | 0],
                                                   * proj_c = attn.proj, proj_x unused (lemevit_amd/ops.py::cstage_pack) */
int lmv_dstage_supported(int C, int heads, int hidden, int H, int W, int M, int dtype);
size_t lmv_dstage_wpk_bytes(int C, int hidden);
size_t lmv_dstage_vec_floats(int C, int hidden);
size_t lmv_dstage_workspace_bytes(int B, int C);
int lmv_dstage_max_concurrent(int C, int H, int kind);   /* lmv_dstage_fwd calls that may be in flight on different streams of the device at once (0: unsupported shape): 8 / 4 at 28 x 28 / 56 x 56,
                                                           * 2 / 1 at 48 x 48 / 96 x 96 -- the caller must not exceed it (lemevit_amd.model falls back to the per-block schedule) */
int lmv_dstage_pack(const lmv_dstage_block_params* p, void* wpk_out, float* vec_out, void* stream);
int lmv_dstage_fwd(const lmv_dstage_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * The stem as ONE launch (csrc/stem.hip; inference form, BatchNorm folded into the convolutions by the caller):
 *   Conv2d(3, Cm, 3, 2, 1) -> BN -> GELU -> Conv2d(Cm, Co, 3, 2, 1) -> BN   (models/lemevit.py:698-704, called at :713)
 * on images x [B, 3, H, W] (element strides sb, sc, sh, sw: NCHW or channels-last, fp32 or bf16) -> y [B, H/4, W/4, Co] bf16 (= the token
 * matrix [B, H/4 * W/4, Co]).  Replaces lmv_im2col3x3s2_c3 + lmv_linear_fwd(GELU) + lmv_im2col3x3s2_nhwc + lmv_linear_fwd: no patch matrix and
 * no Cm-channel map in HBM (an 8 x 8 output tile's 17 x 17 x Cm intermediate lives in LDS).
 *   lmv_stem_supported: (Cm, Co) = (48, 96) (LeMeViT-Base / -Small) or (32, 64) (LeMeViT-Tiny), H and W multiples of 32, bf16 compute.
 *   lmv_stem_pack: w1m [Cm, 32] bf16 (column ci * 9 + ky * 3 + kx, 27..31 zero: the operand of lmv_im2col3x3s2_c3's GEMM), w2m [Co, ld2] bf16
 *     (column (ky * 3 + kx) * Cm + ci: the operand of lmv_im2col3x3s2_nhwc's GEMM) -> wpk_out (lmv_stem_wpk_bytes, MFMA-fragment order).
 *   lmv_stem_fwd: b1 [Cm], b2 [Co] fp32 (the folded biases).
 * ------------------------------------------------------------------------------------------ */
int lmv_stem_supported(int H, int W, int Cm, int Co, int dtype);
size_t lmv_stem_wpk_bytes(int Cm, int Co);
int lmv_stem_pack(const void* w1m, const void* w2m, int ld2, int Cm, int Co, void* wpk_out, void* stream);
int lmv_stem_fwd(const void* x, int x_dtype, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int B, int H, int W, int Cm, int Co, const void* wpk, const float* b1,
                 const float* b2, void* y, void* stream);
/* Residency of the persistent stage kernels.  The workgroups of a slot (the two halves of an image in lmv_sstage_fwd, the image-row workgroups + the meta workgroup of an image in
 * lmv_dstage_fwd) wait for each other inside the launch, so a slot advances only while all of them are resident.  The kernels do NOT rely on a dispatch order or a workgroup -> XCD
 * map for that (HIP promises neither): every workgroup takes a ticket when it starts (csrc/stage_common.h: stage_ticket) and becomes role t % NWG of slot t / NWG, so the started workgroups
 * always form complete slots plus at most 8 incomplete ones per launch, and the next workgroups to start complete those.  Progress needs room for the incomplete slots of ALL launches in
 * flight plus one workgroup: n * 8 (NWG - 1) + 1 <= (workgroups the device holds) -- lmv_sstage_max_concurrent / lmv_dstage_max_concurrent return that n for a shape (launches of
 * different shapes mix under the same per-shape bound), lmv_*stage_supported is 0 on a device too small for n = 1, and the host mirror (lemevit_amd/model.py) takes the per-block
 * schedule where a bound is exceeded.  The bounds count the launches of ONE process; processes sharing a device add up (one process per GPU is the deployment contract).
 * Every in-launch wait is bounded all the same: a spin that runs out (a lost hand-off -- a bug, or more launches in flight than the bound) sets a sticky per-device error word in pinned
 * host memory instead of hanging the GPU.  lmv_stage_error_count returns it without synchronising anything: 0 = every hand-off of every COMPLETED launch arrived (synchronise the streams
 * of interest first for a verdict on them); < 0: LMV_ERR_*; reset != 0 clears it.  lmv_debug_stage_error_set stores a value there from the host (tests of the callers' error paths). */
int lmv_stage_error_count(int reset);
int lmv_debug_stage_error_set(int value);

/* Launch timing probe: lmv_debug_launch_timing(capacity > 0) creates `capacity` event pairs and from then on brackets every lmv_linear_fwd / lmv_linear_res_ln_fwd /
 * lmv_ln_linear_exact_fwd call -- from any schedule, the native block schedule (lmv_block_fwd) included -- with HIP events on the stream it launches on; after a device
 * synchronisation lmv_debug_launch_timing_read returns the number of calls and fills their durations [ms], FLOPs and algorithmic HBM bytes; capacity = 0 frees the events.
 * Not for use inside a stream capture; single host thread.  (bench.py's roofline object.) */
int lmv_debug_launch_timing(int capacity);
int lmv_debug_launch_timing_read(float* ms, double* flops, double* bytes, int* kinds, int capacity);   /* kinds: 0 forward-form Linear, 1 lmv_sstage_fwd, 2 lmv_dstage_fwd, 3 lmv_stem_fwd */
void lmv_stem_debug_timing(void* buf);   /* tools/stem_timeline.py: NULL, or a device buffer of [workgroups][4 waves][8] uint64 s_memtime stamps filled by the next lmv_stem_fwd calls */

#ifdef __cplusplus
}
#endif
#endif /* LEMEVIT_HIP_H */
