// block.hip -- native schedule of one LeMeBlock (models/lemevit.py:500-660) over the kernels of this library: ONE C-ABI call runs the
// 12 (forward) / ~30 (backward) launches of a block, so the host spends ~40 us per block instead of ~25 Python -> ctypes round trips.
//
//   "S" block (:615-650)  forward_with_x : x, c share norm1 / attn / norm2 / mlp weights -> every Linear is one dual-problem launch
//   "D" block (:542-582)  forward_with_xc: dual cross attention, qkv1 / qkv2 and proj_x / proj_c differ
//   "C" block (:584-613)  forward_with_c : only c is updated (cross attention over the image tokens); x is returned untouched
//
// Host code only.  The library still never allocates device memory: every intermediate lives in the caller's `arena` (forward: it doubles
// as the saved-for-backward state, the caller keeps it alive until lmv_block_bwd) or `scratch` (backward; one buffer can serve every
// block of a model: the weight-gradient launches that run on the side stream are joined into the main stream before the call returns).
#include <math.h>
#include <mutex>
#include "common.h"

namespace {

struct Bump {                     // 256-byte aligned bump allocator; base == nullptr: size pass
  unsigned char* base; size_t off, cap;
  void* take(size_t n) {
    const size_t a = (off + 255) & ~(size_t)255;
    off = a + n;
    return base ? (void*)(base + a) : (void*)(uintptr_t)256;      // size pass: any aligned non-null value
  }
};

struct Dims {
  int kind, dtype, B, H, W, N, M, C, Hd, heads;
  size_t es;
  int64_t rows[2];                // image-token rows, meta-token rows
};

int dims_of(const lmv_block_desc* d, Dims* o) {
  if (!d) LMV_FAIL(LMV_ERR_SHAPE, "block: null descriptor");
  if (d->kind < LMV_BLOCK_S || d->kind > LMV_BLOCK_C) LMV_FAIL(LMV_ERR_SHAPE, "block: kind %d (0 = S, 1 = D, 2 = C)", d->kind);
  if (d->dtype != LMV_F32 && d->dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "block: unsupported dtype %d", d->dtype);
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->M <= 0 || d->C <= 0 || (d->C % 32)) LMV_FAIL(LMV_ERR_SHAPE, "block: bad dims B=%d H=%d W=%d M=%d C=%d", d->B, d->H, d->W, d->M, d->C);
  o->kind = d->kind; o->dtype = d->dtype; o->B = d->B; o->H = d->H; o->W = d->W; o->N = d->H * d->W; o->M = d->M; o->C = d->C;
  o->Hd = d->hidden > 0 ? d->hidden : 4 * d->C; o->heads = d->C / 32;
  o->es = d->dtype == LMV_BF16 ? 2 : 4;
  o->rows[0] = (int64_t)d->B * o->N; o->rows[1] = (int64_t)d->B * d->M;
  return LMV_OK;
}

// widths of the packed projections: S / D: qkv of both streams; C: kv of the image tokens, q of the meta tokens
inline int proj_w(const Dims& D, int s) { return D.kind == LMV_BLOCK_C ? (s == 0 ? 2 * D.C : D.C) : 3 * D.C; }

struct Fwd {                      // forward intermediates = saved state of the backward pass ([0] image tokens, [1] meta tokens)
  void *xp, *n1[2], *pj[2], *ao[2], *t2[2], *n2[2], *u[2], *h[2], *ws;
  float *st1[2], *lse[2], *st2[2];
  size_t ws_bytes;
};

// forward attention workspace of `b` images (>= 256), and the slices of lmv_block_fwd_range: the range starting at image i0 takes
// [fwd_ws_off(i0), + fwd_ws_attn(n)); 512 bytes of slack per image keep the 256-byte rounded slices disjoint
inline size_t fwd_ws_attn(const Dims& D, int b) {
  size_t w, w2 = 0;
  if (D.kind == LMV_BLOCK_C) w = lmv_attn_workspace_bytes(b, D.heads, D.M, D.N, 0);
  else if (D.kind == LMV_BLOCK_D) { w = lmv_attn_workspace_bytes(b, D.heads, D.N, D.M, 0); w2 = lmv_attn_workspace_bytes(b, D.heads, D.M, D.N, 0); }
  else { w = lmv_attn_workspace_bytes(b, D.heads, D.N, D.N, 0); w2 = lmv_attn_workspace_bytes(b, D.heads, D.M, D.M, 0); }
  if (w2 > w) w = w2;
  return w < 256 ? 256 : w;
}
inline size_t fwd_ws_off(const Dims& D, int i0) { return i0 <= 0 ? 0 : fwd_ws_attn(D, i0) + (size_t)512 * i0; }
inline size_t fwd_ws_bytes(const Dims& D, int b) { return fwd_ws_attn(D, b) + (size_t)512 * b + 256; }

void layout_fwd(const Dims& D, Bump& a, Fwd* f) {
  const bool cb = D.kind == LMV_BLOCK_C;
  f->xp = a.take(D.rows[0] * D.C * D.es);
  for (int s = 0; s < 2; ++s) {
    f->n1[s] = a.take(D.rows[s] * D.C * D.es);
    f->st1[s] = (float*)a.take(D.rows[s] * 2 * sizeof(float));
    f->pj[s] = a.take(D.rows[s] * proj_w(D, s) * D.es);
    const bool has = !(cb && s == 0);                       // C block: nothing of the image-token stream past the kv projection
    f->ao[s] = has ? a.take(D.rows[s] * D.C * D.es) : nullptr;
    f->lse[s] = has ? (float*)a.take((size_t)D.B * D.heads * (s == 0 ? D.N : D.M) * sizeof(float)) : nullptr;
    f->t2[s] = has ? a.take(D.rows[s] * D.C * D.es) : nullptr;
    f->n2[s] = has ? a.take(D.rows[s] * D.C * D.es) : nullptr;
    f->st2[s] = has ? (float*)a.take(D.rows[s] * 2 * sizeof(float)) : nullptr;
    f->u[s] = has ? a.take(D.rows[s] * D.Hd * D.es) : nullptr;
    f->h[s] = has ? a.take(D.rows[s] * D.Hd * D.es) : nullptr;
  }
  f->ws_bytes = fwd_ws_bytes(D, D.B);
  f->ws = a.take(f->ws_bytes);
}

void attn_desc(lmv_attn_desc* a, const Dims& D, const void* q, int qw, int qo, const void* k, int kw, int ko, const void* v, int vw, int vo,
               void* o, float* lse, int Lq, int Lk, float scale) {
  *a = lmv_attn_desc{};
  a->q = (const unsigned char*)q + qo * D.es; a->k = (const unsigned char*)k + ko * D.es; a->v = (const unsigned char*)v + vo * D.es;
  a->o = o; a->lse = lse;
  a->q_bs = (int64_t)Lq * qw; a->q_rs = qw; a->k_bs = (int64_t)Lk * kw; a->k_rs = kw; a->v_bs = (int64_t)Lk * vw; a->v_rs = vw;
  a->o_bs = (int64_t)Lq * D.C; a->o_rs = D.C;
  a->B = D.B; a->H = D.heads; a->Lq = Lq; a->Lk = Lk; a->scale = scale;
}
void attn_grads(lmv_attn_desc* a, const Dims& D, const void* d_o, void* dq, int qo, void* dk, int ko, void* dv, int vo) {
  a->d_o = d_o;
  a->dq = (unsigned char*)dq + qo * D.es; a->dk = (unsigned char*)dk + ko * D.es; a->dv = (unsigned char*)dv + vo * D.es;
}

lmv_linear_problem prob(const void* a, const void* w, void* out, int64_t rows) {
  lmv_linear_problem p{};
  p.a = a; p.w = w; p.out = out; p.rows = rows;
  return p;
}

#define LMV_TRY(expr) do { const int rc__ = (expr); if (rc__) return rc__; } while (0)

const float SDPA_SCALE = 0.17677669529663687f;        // 32^-1/2 (models/lemevit.py:175: head_dim ** -0.5)
inline void dca_scales(const Dims& D, float* sx, float* sc) {      // models/lemevit.py:235,255-256
  const double base = pow((double)D.C, -0.5);
  *sx = (float)(log((double)D.M) / log((double)D.N) * base);
  *sc = (float)base;
}

// ---- MLP half:  out_s = t_s + ds_s * fc2(GELU(fc1(LN2(t_s))))  for the streams s in [s0, 2) -------------------------------------------
// LMV_BLOCK_FUSED without a backward pass: which Linears take the LayerNorm-folded GEMM (lmv_ln_linear_fwd).  Measured on the Base
// shapes at batch 128 (tools/bench_kernels.py fused): LayerNorm + qkv vs the folded launch is 0.97x at C = 96 (the K = 96 kernel loses
// a resident workgroup to the statistics registers) and 1.0 - 1.08x from C = 192 up.
inline bool fused_on(const lmv_block_desc* d, const Dims& D, int save) {
  return (d->flags & LMV_BLOCK_FUSED) && !save && D.dtype == LMV_BF16 && d->fold_fc1_w && d->fold_fc1_s && d->fold_fc1_b && d->fold_attn_w[0] && d->fold_attn_s[0] &&
         d->fold_attn_b[0] && (D.kind == LMV_BLOCK_S || (d->fold_attn_w[1] && d->fold_attn_s[1] && d->fold_attn_b[1]));
}
inline bool fold_qkv(const Dims& D) { return D.C >= 192; }

// split384 (below): the C = 384 MLP half of the fused inference schedule as three launches instead of the one-kernel form (measured faster
// at 27136 rows -- 9.51 -> 9.38 ms -- and at 37888 rows -- Base 384^2 forward 16.5 -> 16.1 ms)
inline bool mlp_split384(const Dims& D) { const int64_t r = D.rows[0] + D.rows[1]; return lmv_config().mlp_split384 && D.C == 384 && D.Hd == 1536 && r >= 16384 && r <= 65536; }
// "S" blocks: the attention projection (residual epilogue) and the norm2 that follows it in ONE launch (lmv_linear_res_ln_fwd), wherever
// the MLP half reads norm2's output from memory (training; the split inference form) -- lemevit_amd/ops.py::res_ln_fused is the same rule
inline bool res_ln_ok(const lmv_block_desc* d, const Dims& D, int save) {
  if (D.kind != LMV_BLOCK_S || D.dtype != LMV_BF16 || !lmv_config().res_ln_fused || D.rows[0] + D.rows[1] < 16384 || D.rows[0] + D.rows[1] > 32768) return false;
  if (!lmv_linear_res_ln_fwd_supported(D.C, D.C, D.dtype)) return false;
  return !(fused_on(d, D, save) && !mlp_split384(D));
}

int mlp_fwd(const lmv_block_desc* d, const Dims& D, const Fwd& f, int s0, void* const* outs, const float* const* ds, int save, void* st, bool ln_done = false) {
  const int ns = 2 - s0;
  // C = 384 with enough rows to fill the chip: LayerNorm + register-stationary fc1 (csrc/rsgemm.hip) + whole-width fc2 (csrc/wngemm.hip)
  // beat the one-kernel MLP (one 160 KB workgroup per CU at this width): Base 224 forward pass 9.51 -> 9.38 ms (LMV_MLP_SPLIT384=0: A/B)
  const bool split384 = mlp_split384(D);
  if (fused_on(d, D, save) && !split384) {
    if (lmv_mlp_fused_supported(D.C, D.Hd, D.dtype)) {      // LN2 -> fc1 -> GELU -> fc2 -> + residual: one kernel, hidden on chip
      lmv_mlp_problem q[2] = {};
      for (int i = 0; i < ns; ++i) { const int s = s0 + i; q[i].x = f.t2[s]; q[i].out = outs[s]; q[i].row_scale = ds[s]; q[i].rows = D.rows[s]; q[i].rows_per_sample = s == 0 ? D.N : D.M; }
      lmv_mlp_weights w{d->fold_fc1_w, d->fold_fc1_s, d->fold_fc1_b, d->fc2_w, d->fc2_b};
      return lmv_mlp_fused_fwd(q, ns, &w, D.C, D.Hd, d->eps, D.dtype, st);
    }
    lmv_linear_problem p[2];                                 // (C = 512: LN2 folded into fc1, then fc2)
    for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(f.t2[s], d->fold_fc1_w, f.h[s], D.rows[s]); p[i].bias = d->fold_fc1_b; p[i].aux = d->fold_fc1_s; }
    LMV_TRY(lmv_ln_linear_fwd(p, ns, D.Hd, D.C, d->eps, LMV_ACT_GELU, D.dtype, st));
    for (int i = 0; i < ns; ++i) {
      const int s = s0 + i;
      p[i] = prob(f.h[s], d->fc2_w, outs[s], D.rows[s]); p[i].bias = d->fc2_b; p[i].res = f.t2[s]; p[i].row_scale = ds[s];
      p[i].rows_per_sample = s == 0 ? D.N : D.M;
    }
    return lmv_linear_fwd(p, ns, D.C, D.Hd, LMV_ACT_NONE, D.dtype, st);
  }
  if (!ln_done) {
    lmv_ln_segment seg[2] = {};
    for (int i = 0; i < ns; ++i) { const int s = s0 + i; seg[i].x = f.t2[s]; seg[i].y = f.n2[s]; seg[i].stats = save ? f.st2[s] : nullptr; seg[i].rows = D.rows[s]; }
    LMV_TRY(lmv_layernorm_fwd(seg, ns, d->n2_w, d->n2_b, D.C, d->eps, D.dtype, st));
  }
  lmv_linear_problem p[2];
  for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(f.n2[s], d->fc1_w, f.h[s], D.rows[s]); p[i].bias = d->fc1_b; p[i].out_pre = save ? f.u[s] : nullptr; }
  LMV_TRY(lmv_linear_fwd(p, ns, D.Hd, D.C, LMV_ACT_GELU, D.dtype, st));
  for (int i = 0; i < ns; ++i) {
    const int s = s0 + i;
    p[i] = prob(f.h[s], d->fc2_w, outs[s], D.rows[s]); p[i].bias = d->fc2_b; p[i].res = f.t2[s]; p[i].row_scale = ds[s];
    p[i].rows_per_sample = s == 0 ? D.N : D.M;
  }
  return lmv_linear_fwd(p, ns, D.C, D.Hd, LMV_ACT_NONE, D.dtype, st);
}

// ---- backward plumbing: weight-gradient launches go to the side stream ---------------------------------------------------------------
struct Side {
  hipStream_t main, side; hipEvent_t fork, join; bool used;
  void* ws; size_t ws_bytes;
  // The small column reductions of a block -- LayerNorm dgamma | dbeta rows of norm2 and norm1, the depth-wise convolution's tap sums --
  // are DEFERRED: each keeps its own partial-sum buffer, describes it in `segs`, and ONE lmv_reduce_batch launch at the end of the
  // block sums them all (same summation trees: bit-identical to the per-op reduces; 3 launches of ~5 us per block become one).
  // The split-K slabs of the weight-gradient GEMMs are NOT deferred: summed right behind their GEMM they are read back out of the
  // 256 MB MALL; kept until the end of the block (165 MB per stage-3 block) they go to HBM and back -- measured twice: +0.5 (round 2)
  // and +0.8 ms per train step (round 3).
  void* ws_tail; size_t ws_tail_bytes;         // region of the side workspace behind the weight-gradient slabs (dwconv tap sums)
  lmv_reduce_seg segs[LMV_REDUCE_MAX_SEGS]; int nsegs = 0;
  hipStream_t begin() {           // the stream a weight-gradient launch goes to, made to wait for everything enqueued on `main` so far
    if (!side) return main;
    (void)hipEventRecord(fork, main);
    (void)hipStreamWaitEvent(side, fork, 0);
    used = true;
    return side;
  }
  void finish() {
    if (side && used) { (void)hipEventRecord(join, side); (void)hipStreamWaitEvent(main, join, 0); used = false; }
  }
};

// Fork / join events of ONE lmv_block_bwd call.  They come from a per-device free list and are owned exclusively by the call until it
// returns, so two host threads (autograd's per-device backward threads under DDP, or two caller threads on two streams) can run
// lmv_block_bwd on one device at the same time: with ONE shared pair, thread A's hipStreamWaitEvent(side, fork) could pick up the
// hipEventRecord(fork, ...) thread B had just made on ITS stream.  An event that is handed back may still be referenced by a wait that has
// not executed yet: a stream wait binds to the record that was current when the wait was ENQUEUED, so re-recording it later is safe.
struct EventPool {
  std::mutex mu;
  hipEvent_t free_ev[64][32];
  int nfree[64] = {};
  int take(hipEvent_t* fork, hipEvent_t* join) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    hipEvent_t out[2];
    int got = 0;
    {
      std::lock_guard<std::mutex> lk(mu);
      while (got < 2 && nfree[dev] > 0) out[got++] = free_ev[dev][--nfree[dev]];
    }
    for (; got < 2; ++got)
      if (hipEventCreateWithFlags(&out[got], hipEventDisableTiming) != hipSuccess) {
        give(dev, out, got);
        LMV_FAIL(LMV_ERR_LAUNCH, "block_bwd: cannot create stream events");
      }
    *fork = out[0]; *join = out[1];
    return LMV_OK;
  }
  void give(int dev, const hipEvent_t* ev, int n) {
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < n; ++i) {
      if (nfree[dev] < 32) free_ev[dev][nfree[dev]++] = ev[i];
      else (void)hipEventDestroy(ev[i]);
    }
  }
  void give(hipEvent_t fork, hipEvent_t join) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const hipEvent_t ev[2] = {fork, join};
    give(dev & 63, ev, 2);
  }
};
EventPool& event_pool() { static EventPool p; return p; }

int dw(Side& sd, const lmv_linear_problem* p, int np, int N, int K, int dtype) {
  const size_t need = lmv_linear_dw_workspace_bytes(p, np, N, K, dtype);
  if (need > sd.ws_bytes) LMV_FAIL(LMV_ERR_WORKSPACE, "block_bwd: weight-gradient workspace %zu > %zu bytes", need, sd.ws_bytes);
  return lmv_linear_dw(p, np, N, K, sd.ws, sd.ws_bytes, dtype, sd.begin());
}
// every deferred reduction of the block in one launch, on the side stream behind the launches that produced the partial sums
int flush_reduces(Side& sd) {
  if (!sd.nsegs) return LMV_OK;
  const int rc = lmv_reduce_batch(sd.segs, sd.nsegs, sd.begin());
  sd.nsegs = 0;
  return rc;
}

struct Bwd {                      // backward temporaries ([0] image tokens, [1] meta tokens)
  void *g[2], *du[2], *dn2[2], *dt2[2], *g2[2], *dao[2], *dpj[2], *dn1[2], *dxp, *ws_main, *ws_side, *ws_conv, *ws_ln[2];
  size_t ws_main_bytes, ws_side_bytes, ws_conv_bytes, ws_ln_bytes;
};

size_t max_dw_ws(const Dims& D) {
  // the weight-gradient GEMMs of a block: (N, K) per layer, rows = image + meta tokens (shared weights) or one stream each
  const void* fake = (const void*)(uintptr_t)256;
  size_t m = 256;
  auto one = [&](int np, int64_t r0, int64_t r1, int N, int K, bool shared) {
    lmv_linear_problem p[2];
    float* o0 = (float*)(uintptr_t)256; float* o1 = shared ? o0 : (float*)(uintptr_t)512;
    p[0] = prob(fake, fake, o0, r0); p[0].bias_grad = (float*)(uintptr_t)1024;
    p[1] = prob(fake, fake, o1, r1); p[1].bias_grad = shared ? p[0].bias_grad : (float*)(uintptr_t)2048;
    const size_t w = lmv_linear_dw_workspace_bytes(p, np, N, K, D.dtype);
    if (w > m) m = w;
  };
  const int C = D.C, Hd = D.Hd;
  if (D.kind == LMV_BLOCK_C) {
    one(1, D.rows[1], 0, C, Hd, true); one(1, D.rows[1], 0, Hd, C, true);                                           // fc2, fc1 (meta tokens)
    one(1, D.rows[1], 0, C, C, true); one(1, D.rows[1], 0, C, C, true); one(1, D.rows[0], 0, 2 * C, C, true);      // proj, q (meta tokens), kv (image tokens)
  } else {
    const bool sh = D.kind == LMV_BLOCK_S;
    one(2, D.rows[0], D.rows[1], C, Hd, true); one(2, D.rows[0], D.rows[1], Hd, C, true);
    one(2, D.rows[0], D.rows[1], C, C, sh); one(2, D.rows[0], D.rows[1], 3 * C, C, sh);
  }
  return (m + 255) & ~(size_t)255;
}

void layout_bwd(const Dims& D, Bump& a, Bwd* b) {
  const bool cb = D.kind == LMV_BLOCK_C;
  for (int s = 0; s < 2; ++s) {
    const bool has = !(cb && s == 0);
    b->g[s] = has ? a.take(D.rows[s] * D.C * D.es) : nullptr;
    b->du[s] = has ? a.take(D.rows[s] * D.Hd * D.es) : nullptr;
    b->dn2[s] = has ? a.take(D.rows[s] * D.C * D.es) : nullptr;
    b->dt2[s] = has ? a.take(D.rows[s] * D.C * D.es) : nullptr;
    b->g2[s] = has ? a.take(D.rows[s] * D.C * D.es) : nullptr;
    b->dao[s] = has ? a.take(D.rows[s] * D.C * D.es) : nullptr;
    b->dpj[s] = a.take(D.rows[s] * proj_w(D, s) * D.es);
    b->dn1[s] = a.take(D.rows[s] * D.C * D.es);
  }
  b->dxp = a.take(D.rows[0] * D.C * D.es);
  size_t w = lmv_layernorm_bwd_workspace_bytes(D.rows[0] + D.rows[1], D.C, D.dtype);
  size_t w2;
  if (cb) w2 = lmv_attn_workspace_bytes(D.B, D.heads, D.M, D.N, 1);
  else if (D.kind == LMV_BLOCK_D) { w2 = lmv_attn_workspace_bytes(D.B, D.heads, D.N, D.M, 1); const size_t w3 = lmv_attn_workspace_bytes(D.B, D.heads, D.M, D.N, 1); if (w3 > w2) w2 = w3; }
  else { w2 = lmv_attn_workspace_bytes(D.B, D.heads, D.N, D.N, 1); const size_t w3 = lmv_attn_workspace_bytes(D.B, D.heads, D.M, D.M, 1); if (w3 > w2) w2 = w3; }
  if (w2 > w) w = w2;
  if (w < 256) w = 256;
  b->ws_main_bytes = w; b->ws_main = a.take(w);
  b->ws_side_bytes = max_dw_ws(D); b->ws_side = a.take(b->ws_side_bytes);
  b->ws_conv_bytes = lmv_dwconv3x3_bwd_weight_workspace_bytes(D.B, D.H, D.W, D.C, D.dtype); b->ws_conv = a.take(b->ws_conv_bytes);
  // LayerNorm dgamma / dbeta partial rows of norm2 and norm1: their reduces run on the side stream, so each keeps its own buffer
  b->ws_ln_bytes = lmv_layernorm_bwd_workspace_bytes(D.rows[0] + D.rows[1], D.C, D.dtype);
  { const size_t w2 = lmv_linear_dx_ln_bwd_workspace_bytes(D.rows[0] + D.rows[1], D.C); if (w2 > b->ws_ln_bytes) b->ws_ln_bytes = w2; }      // (the fused dX + LayerNorm form: one row per 128-row panel)
  for (int i = 0; i < 2; ++i) b->ws_ln[i] = a.take(b->ws_ln_bytes);
}

// LayerNorm backward with the dgamma / dbeta reduce off the critical path: dx on the main stream, the ~5 us reduce launch behind a
// fork on the weight-gradient side stream (in line when there is none)
int ln_bwd(Side& sd, const lmv_ln_segment* seg, int nseg, const float* gamma, float* dgamma, float* dbeta, const Dims& D, void* ws, size_t ws_bytes) {
  int rows = 0;
  LMV_TRY(lmv_layernorm_bwd_partial(seg, nseg, gamma, D.C, ws, ws_bytes, &rows, D.dtype, sd.main));
  if (sd.nsegs + 1 > LMV_REDUCE_MAX_SEGS) LMV_FAIL(LMV_ERR_WORKSPACE, "block_bwd: too many deferred reductions");
  lmv_reduce_seg& sg = sd.segs[sd.nsegs++];      // (dgamma | dbeta) rows of the per-workgroup partial sums: summed by the block's reduce launch
  sg = lmv_reduce_seg{};
  sg.ws = (const float*)ws; sg.out_w = dgamma; sg.out_b = dbeta; sg.slab_stride = 2 * D.C; sg.nw = D.C; sg.nslabs = rows; sg.nb = D.C;
  sg.kind = LMV_REDUCE_ROWS; sg.mode = 0;
  return LMV_OK;
}

// The same LayerNorm backward fused into the dX GEMM that produces its dy (csrc/wngemm.hip: C = 384, bf16, transposed weight copy present,
// enough rows to fill the chip): dy never reaches memory.  p[i].a = dY, p[i].w = the transposed weight; seg as for ln_bwd.
inline bool dx_ln_fused_ok(const Dims& D, const void* wt, int N) {
  const int64_t rows = D.rows[0] + D.rows[1];      // one round of 128-row panels on the chip (csrc/wngemm.hip::lmv_wn_eligible)
  return wt && D.dtype == LMV_BF16 && lmv_config().dx_ln_fused && lmv_linear_dx_ln_bwd_supported(D.C, N, D.dtype) && rows >= 16384 && rows <= 32768;
}
int dx_ln_bwd(Side& sd, const lmv_linear_problem* p, const lmv_ln_segment* seg, int nseg, int N, const float* gamma, float* dgamma, float* dbeta, const Dims& D,
              void* ws, size_t ws_bytes) {
  int rows = 0;
  LMV_TRY(lmv_linear_dx_ln_bwd(p, seg, nseg, D.C, N, gamma, ws, ws_bytes, &rows, D.dtype, sd.main));
  if (sd.nsegs + 1 > LMV_REDUCE_MAX_SEGS) LMV_FAIL(LMV_ERR_WORKSPACE, "block_bwd: too many deferred reductions");
  lmv_reduce_seg& sg = sd.segs[sd.nsegs++];
  sg = lmv_reduce_seg{};
  sg.ws = (const float*)ws; sg.out_w = dgamma; sg.out_b = dbeta; sg.slab_stride = 2 * D.C; sg.nw = D.C; sg.nslabs = rows; sg.nb = D.C;
  sg.kind = LMV_REDUCE_ROWS; sg.mode = 0;
  return LMV_OK;
}

// MLP half backward (blocks.py::_mlp_bwd): douts = gradients of the block outputs, returns dt2 (gradient of the MLP half's input) and, where the
// attention half's DropPath vector nds[s] is set, g2[s] = dt2[s] pre-scaled by it (written by the same LayerNorm-backward launch)
int mlp_bwd(const lmv_block_desc* d, const Dims& D, const Fwd& f, const Bwd& b, int s0, const void* const* douts, const float* const* ds, const float* const* nds,
            const void** g2_out, Side& sd) {
  const int ns = 2 - s0;
  hipStream_t st = sd.main;
  const void* g[2];
  lmv_row_scale_segment rs[2]; int nrs = 0;
  for (int s = s0; s < 2; ++s) {
    g[s] = douts[s];
    if (ds[s]) { rs[nrs].x = douts[s]; rs[nrs].scale = ds[s]; rs[nrs].y = b.g[s]; rs[nrs].rows = D.rows[s]; rs[nrs].rows_per_sample = s == 0 ? D.N : D.M; ++nrs; g[s] = b.g[s]; }
  }
  if (nrs) LMV_TRY(lmv_row_scale_multi(rs, nrs, D.C, D.dtype, st));
  lmv_linear_problem p[2];
  for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(g[s], f.h[s], d->g_fc2_w, D.rows[s]); p[i].bias_grad = d->g_fc2_b; }
  LMV_TRY(dw(sd, p, ns, D.C, D.Hd, D.dtype));
  if (d->fc2_wt && D.dtype == LMV_BF16) {
    // dX of fc2 as a forward-form GEMM on the transposed weight copy [hidden, C]: du = (g W2) * GELU'(u) = (g . fc2_wt^T) * GELU'(u)
    // (the register-stationary kernel takes it for C = 192 / 384: csrc/rsgemm.hip)
    for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(g[s], d->fc2_wt, b.du[s], D.rows[s]); p[i].aux = f.u[s]; }
    LMV_TRY(lmv_linear_fwd(p, ns, D.Hd, D.C, LMV_ACT_GELU_GRAD, D.dtype, st));
  } else {
    for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(g[s], d->fc2_w, b.du[s], D.rows[s]); p[i].aux = f.u[s]; }
    LMV_TRY(lmv_linear_dx(p, ns, D.C, D.Hd, LMV_ACT_GELU_GRAD, D.dtype, st));
  }
  for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(b.du[s], f.n2[s], d->g_fc1_w, D.rows[s]); p[i].bias_grad = d->g_fc1_b; }
  LMV_TRY(dw(sd, p, ns, D.Hd, D.C, D.dtype));
  const bool fuse2 = dx_ln_fused_ok(D, d->fc1_wt, D.Hd);      // dX of fc1 + LayerNorm-2 backward in one kernel
  if (fuse2) {
  } else if (d->fc1_wt && D.dtype == LMV_BF16) {      // dX of fc1 as a forward-form GEMM on the transposed weight copy [C, hidden] (csrc/wngemm.hip at C = 384)
    for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(b.du[s], d->fc1_wt, b.dn2[s], D.rows[s]); }
    LMV_TRY(lmv_linear_fwd(p, ns, D.C, D.Hd, LMV_ACT_NONE, D.dtype, st));
  } else {
    for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(b.du[s], d->fc1_w, b.dn2[s], D.rows[s]); }
    LMV_TRY(lmv_linear_dx(p, ns, D.Hd, D.C, LMV_ACT_NONE, D.dtype, st));
  }
  lmv_ln_segment seg[2] = {};
  for (int i = 0; i < ns; ++i) {
    const int s = s0 + i;
    seg[i].x = f.t2[s]; seg[i].dy = b.dn2[s]; seg[i].stats = f.st2[s]; seg[i].dres = douts[s]; seg[i].dx = b.dt2[s]; seg[i].rows = D.rows[s];
    g2_out[s] = b.dt2[s];
    if (nds && nds[s]) { seg[i].dx_scale = nds[s]; seg[i].dx_scaled = b.g2[s]; seg[i].rows_per_sample = s == 0 ? D.N : D.M; g2_out[s] = b.g2[s]; }
  }
  if (fuse2) {
    for (int i = 0; i < ns; ++i) { const int s = s0 + i; p[i] = prob(b.du[s], d->fc1_wt, b.dn2[s], D.rows[s]); }
    return dx_ln_bwd(sd, p, seg, ns, D.Hd, d->n2_w, d->g_n2_w, d->g_n2_b, D, b.ws_ln[0], b.ws_ln_bytes);
  }
  return ln_bwd(sd, seg, ns, d->n2_w, d->g_n2_w, d->g_n2_b, D, b.ws_ln[0], b.ws_ln_bytes);
}

int check_ptrs(const lmv_block_desc* d, bool grads) {
  const void* need[] = {d->pos_w, d->pos_b, d->n1_w, d->n1_b, d->attn_w[0], d->attn_b[0], d->attn_w[1], d->attn_b[1], d->n2_w, d->n2_b, d->fc1_w, d->fc1_b, d->fc2_w, d->fc2_b};
  for (const void* p : need) if (!p) LMV_FAIL(LMV_ERR_SHAPE, "block: null parameter pointer");
  if (d->kind != LMV_BLOCK_S && (!d->attn_w[2] || !d->attn_b[2])) LMV_FAIL(LMV_ERR_SHAPE, "block: null parameter pointer");
  if (d->kind == LMV_BLOCK_D && (!d->attn_w[3] || !d->attn_b[3])) LMV_FAIL(LMV_ERR_SHAPE, "block: null parameter pointer");
  if (grads) {
    const void* gn[] = {d->g_pos_w, d->g_pos_b, d->g_n1_w, d->g_n1_b, d->g_attn_w[0], d->g_attn_b[0], d->g_attn_w[1], d->g_attn_b[1], d->g_n2_w, d->g_n2_b, d->g_fc1_w, d->g_fc1_b,
                        d->g_fc2_w, d->g_fc2_b};
    for (const void* p : gn) if (!p) LMV_FAIL(LMV_ERR_SHAPE, "block_bwd: null gradient pointer");
    if (d->kind != LMV_BLOCK_S && (!d->g_attn_w[2] || !d->g_attn_b[2])) LMV_FAIL(LMV_ERR_SHAPE, "block_bwd: null gradient pointer");
    if (d->kind == LMV_BLOCK_D && (!d->g_attn_w[3] || !d->g_attn_b[3])) LMV_FAIL(LMV_ERR_SHAPE, "block_bwd: null gradient pointer");
  }
  return LMV_OK;
}

}  // namespace

extern "C" size_t lmv_block_arena_bytes(const lmv_block_desc* d) {
  Dims D;
  if (dims_of(d, &D)) return 0;
  Bump a{nullptr, 0, 0};
  Fwd f;
  layout_fwd(D, a, &f);
  return a.off + 256;
}

extern "C" size_t lmv_block_bwd_scratch_bytes(const lmv_block_desc* d) {
  Dims D;
  if (dims_of(d, &D)) return 0;
  Bump a{nullptr, 0, 0};
  Bwd b;
  layout_bwd(D, a, &b);
  return a.off + 256;
}

namespace {
int block_fwd_body(const lmv_block_desc* d, const Dims& D, const Fwd& f, const void* x, const void* c, void* x_out, void* c_out, int save, void* stream);
}  // namespace

extern "C" int lmv_block_fwd_range(const lmv_block_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* arena, size_t arena_bytes, int save,
                                   int image0, int nimages, void* stream) {
  Dims DF;
  LMV_TRY(dims_of(d, &DF));
  LMV_TRY(check_ptrs(d, false));
  const bool cb = DF.kind == LMV_BLOCK_C;
  if (!x || !c || !c_out || (!cb && !x_out) || !arena || !lmv_aligned16(arena)) LMV_FAIL(LMV_ERR_SHAPE, "block_fwd: null / misaligned tensor");
  if (image0 < 0 || nimages <= 0 || image0 + nimages > DF.B) LMV_FAIL(LMV_ERR_SHAPE, "block_fwd: images [%d, %d) outside the batch of %d", image0, image0 + nimages, DF.B);
  Bump a{(unsigned char*)arena, 0, arena_bytes};
  Fwd f;
  layout_fwd(DF, a, &f);
  if (a.off > arena_bytes) LMV_FAIL(LMV_ERR_WORKSPACE, "block_fwd: arena %zu < %zu bytes", arena_bytes, a.off);
  if (image0 == 0 && nimages == DF.B) return block_fwd_body(d, DF, f, x, c, x_out, c_out, save, stream);
  // a range of images: every tensor of the arena is [rows, width] with the images outermost, so the range is a contiguous slice of each
  lmv_block_desc dr = *d;
  dr.B = nimages;
  for (int k = 0; k < 4; ++k) if (dr.masks[k]) dr.masks[k] += image0;
  Dims D;
  LMV_TRY(dims_of(&dr, &D));
  const size_t es = D.es;
  auto adv = [&](void* ptr, int s, size_t width_bytes) -> void* { return ptr ? (void*)((unsigned char*)ptr + (size_t)image0 * (s == 0 ? D.N : D.M) * width_bytes) : nullptr; };
  f.xp = adv(f.xp, 0, D.C * es);
  for (int s = 0; s < 2; ++s) {
    f.n1[s] = adv(f.n1[s], s, D.C * es); f.st1[s] = (float*)adv(f.st1[s], s, 2 * sizeof(float)); f.pj[s] = adv(f.pj[s], s, proj_w(D, s) * es);
    f.ao[s] = adv(f.ao[s], s, D.C * es); f.lse[s] = (float*)adv(f.lse[s], s, D.heads * sizeof(float)); f.t2[s] = adv(f.t2[s], s, D.C * es);
    f.n2[s] = adv(f.n2[s], s, D.C * es); f.st2[s] = (float*)adv(f.st2[s], s, 2 * sizeof(float)); f.u[s] = adv(f.u[s], s, D.Hd * es); f.h[s] = adv(f.h[s], s, D.Hd * es);
  }
  const size_t w0 = fwd_ws_off(DF, image0), w1 = fwd_ws_attn(DF, nimages);
  if (w0 + w1 > f.ws_bytes) LMV_FAIL(LMV_ERR_WORKSPACE, "block_fwd: attention workspace slice [%zu, %zu) outside %zu bytes", w0, w0 + w1, f.ws_bytes);
  f.ws = (unsigned char*)f.ws + w0; f.ws_bytes = w1;
  return block_fwd_body(&dr, D, f, adv(const_cast<void*>(x), 0, D.C * es), adv(const_cast<void*>(c), 1, D.C * es), adv(x_out, 0, D.C * es), adv(c_out, 1, D.C * es), save, stream);
}

extern "C" int lmv_block_fwd(const lmv_block_desc* d, const void* x, const void* c, void* x_out, void* c_out, void* arena, size_t arena_bytes, int save, void* stream) {
  return lmv_block_fwd_range(d, x, c, x_out, c_out, arena, arena_bytes, save, 0, d ? d->B : 0, stream);
}

namespace {

int block_fwd_body(const lmv_block_desc* d, const Dims& D, const Fwd& f, const void* x, const void* c, void* x_out, void* c_out, int save, void* stream) {
  const bool cb = D.kind == LMV_BLOCK_C;
  const int C = D.C, N = D.N, M = D.M;
  LMV_TRY(lmv_dwconv3x3_residual_fwd(x, d->pos_w, d->pos_b, f.xp, D.B, D.H, D.W, C, D.dtype, stream));          // :546
  const bool fq = fused_on(d, D, save) && fold_qkv(D);      // norm1 folded into the projections: no LayerNorm launch, no normalised copy
  // C = 96 blocks: norm1 runs INSIDE the projection launches (csrc/rswgemm.hip: exact LayerNorm on the register-resident rows, the
  // normalised rows and their statistics written from there) -- lemevit_amd/ops.py::ln_exact_fused is the same rule
  const bool ex = !fq && D.dtype == LMV_BF16 && lmv_config().ln_exact_fused && D.kind != LMV_BLOCK_S &&
                  lmv_ln_linear_exact_fwd_supported(cb ? 2 * D.C : 3 * D.C, D.C, D.dtype) && lmv_ln_linear_exact_fwd_supported(D.C, D.C, D.dtype);
  const void* src[2] = {f.xp, c};                           // token rows the projections read (fq / ex: raw; else LN1 output)
  if (!fq && !ex) {
    lmv_ln_segment seg[2] = {};
    seg[0].x = f.xp; seg[0].y = f.n1[0]; seg[0].stats = save ? f.st1[0] : nullptr; seg[0].rows = D.rows[0];
    seg[1].x = c; seg[1].y = f.n1[1]; seg[1].stats = save ? f.st1[1] : nullptr; seg[1].rows = D.rows[1];
    LMV_TRY(lmv_layernorm_fwd(seg, 2, d->n1_w, d->n1_b, C, d->eps, D.dtype, stream));
    src[0] = f.n1[0]; src[1] = f.n1[1];
  }
  // one projection launch: stream s through attention weight slot `slot[s]` (folded or plain operands)
  auto project = [&](lmv_linear_problem* p, int np, const int* streams, const int* slots, int width) -> int {
    for (int i = 0; i < np; ++i) {
      const int s = streams[i], k = slots[i];
      p[i] = prob(src[s], fq ? d->fold_attn_w[k] : d->attn_w[k], f.pj[s], D.rows[s]);
      p[i].bias = fq ? d->fold_attn_b[k] : d->attn_b[k];
      if (fq) p[i].aux = d->fold_attn_s[k];
    }
    if (ex) {
      lmv_ln_segment seg[2] = {};
      for (int i = 0; i < np; ++i) { const int s = streams[i]; seg[i].y = save ? f.n1[s] : nullptr; seg[i].stats = save ? f.st1[s] : nullptr; seg[i].rows = D.rows[s]; }      // inference: the normalised rows never leave the registers
      return lmv_ln_linear_exact_fwd(p, seg, np, width, C, d->n1_w, d->n1_b, d->eps, D.dtype, stream);
    }
    return fq ? lmv_ln_linear_fwd(p, np, width, C, d->eps, LMV_ACT_NONE, D.dtype, stream) : lmv_linear_fwd(p, np, width, C, LMV_ACT_NONE, D.dtype, stream);
  };
  lmv_linear_problem p[2];
  lmv_attn_desc ad[2];
  void* outs[2] = {x_out, c_out};
  bool ln2_done = false;
  if (D.kind == LMV_BLOCK_S) {                    // the SAME weights for x and c (:632,634)
    { const int st_[2] = {0, 1}, sl_[2] = {0, 0}; LMV_TRY(project(p, 2, st_, sl_, 3 * C)); }
    for (int s = 0; s < 2; ++s) { const int L = s == 0 ? N : M; attn_desc(&ad[s], D, f.pj[s], 3 * C, 0, f.pj[s], 3 * C, C, f.pj[s], 3 * C, 2 * C, f.ao[s], save ? f.lse[s] : nullptr, L, L, SDPA_SCALE); }
    LMV_TRY(lmv_attn_fwd_pair(ad, f.ws, f.ws_bytes, D.dtype, stream));
    for (int s = 0; s < 2; ++s) {
      p[s] = prob(f.ao[s], d->attn_w[1], f.t2[s], D.rows[s]); p[s].bias = d->attn_b[1]; p[s].res = s == 0 ? f.xp : c;
      p[s].row_scale = d->masks[s == 0 ? 0 : 2]; p[s].rows_per_sample = s == 0 ? N : M;
    }
    if (res_ln_ok(d, D, save)) {              // ... together with norm2 (csrc/wngemm.hip)
      lmv_ln_segment seg[2] = {};
      for (int s = 0; s < 2; ++s) { seg[s].y = f.n2[s]; seg[s].stats = save ? f.st2[s] : nullptr; seg[s].rows = D.rows[s]; }
      LMV_TRY(lmv_linear_res_ln_fwd(p, seg, 2, C, C, d->n2_w, d->n2_b, d->eps, D.dtype, stream));
      ln2_done = true;
    } else {
      LMV_TRY(lmv_linear_fwd(p, 2, C, C, LMV_ACT_NONE, D.dtype, stream));
    }
  } else if (D.kind == LMV_BLOCK_D) {             // :288-302
    float sx, sc;
    dca_scales(D, &sx, &sc);
    { const int st_[2] = {0, 1}, sl_[2] = {0, 1}; LMV_TRY(project(p, 2, st_, sl_, 3 * C)); }
    attn_desc(&ad[0], D, f.pj[0], 3 * C, 0, f.pj[1], 3 * C, C, f.pj[1], 3 * C, 2 * C, f.ao[0], save ? f.lse[0] : nullptr, N, M, sx);      // image -> meta (:297)
    attn_desc(&ad[1], D, f.pj[1], 3 * C, 0, f.pj[0], 3 * C, C, f.pj[0], 3 * C, 2 * C, f.ao[1], save ? f.lse[1] : nullptr, M, N, sc);      // meta -> image (:300)
    LMV_TRY(lmv_attn_fwd(&ad[0], f.ws, f.ws_bytes, D.dtype, stream));
    LMV_TRY(lmv_attn_fwd(&ad[1], f.ws, f.ws_bytes, D.dtype, stream));
    for (int s = 0; s < 2; ++s) {
      p[s] = prob(f.ao[s], d->attn_w[2 + s], f.t2[s], D.rows[s]); p[s].bias = d->attn_b[2 + s]; p[s].res = s == 0 ? f.xp : c;
      p[s].row_scale = d->masks[s == 0 ? 0 : 2]; p[s].rows_per_sample = s == 0 ? N : M;
    }
    LMV_TRY(lmv_linear_fwd(p, 2, C, C, LMV_ACT_NONE, D.dtype, stream));
  } else {                                        // :477-486,600  q = attn_w[0] (meta tokens), kv = attn_w[1] (image tokens), proj = attn_w[2]
    { const int st_[1] = {0}, sl_[1] = {1}; LMV_TRY(project(p, 1, st_, sl_, 2 * C)); }
    { const int st_[1] = {1}, sl_[1] = {0}; LMV_TRY(project(p, 1, st_, sl_, C)); }
    attn_desc(&ad[0], D, f.pj[1], C, 0, f.pj[0], 2 * C, 0, f.pj[0], 2 * C, C, f.ao[1], save ? f.lse[1] : nullptr, M, N, SDPA_SCALE);
    LMV_TRY(lmv_attn_fwd(&ad[0], f.ws, f.ws_bytes, D.dtype, stream));
    p[0] = prob(f.ao[1], d->attn_w[2], f.t2[1], D.rows[1]); p[0].bias = d->attn_b[2]; p[0].res = c; p[0].row_scale = d->masks[0]; p[0].rows_per_sample = M;
    LMV_TRY(lmv_linear_fwd(p, 1, C, C, LMV_ACT_NONE, D.dtype, stream));
  }
  const float* ds[2] = {cb ? nullptr : d->masks[1], cb ? d->masks[1] : d->masks[3]};
  return mlp_fwd(d, D, f, cb ? 1 : 0, outs, ds, save, stream, ln2_done);
}

}  // namespace

extern "C" int lmv_block_bwd(const lmv_block_desc* d, const void* x, const void* c, const void* arena, size_t arena_bytes, const void* dx_out, const void* dc_out, void* dx, void* dc,
                             void* scratch, size_t scratch_bytes, void* stream, void* side_stream) {
  Dims D;
  LMV_TRY(dims_of(d, &D));
  LMV_TRY(check_ptrs(d, true));
  const bool cb = D.kind == LMV_BLOCK_C;
  if (!x || !c || !arena || !dc_out || (!cb && !dx_out) || !dx || !dc || !scratch || !lmv_aligned16(scratch)) LMV_FAIL(LMV_ERR_SHAPE, "block_bwd: null / misaligned tensor");
  Bump a{(unsigned char*)const_cast<void*>(arena), 0, arena_bytes};
  Fwd f;
  layout_fwd(D, a, &f);
  if (a.off > arena_bytes) LMV_FAIL(LMV_ERR_WORKSPACE, "block_bwd: arena %zu < %zu bytes", arena_bytes, a.off);
  Bump s{(unsigned char*)scratch, 0, scratch_bytes};
  Bwd b;
  layout_bwd(D, s, &b);
  if (s.off > scratch_bytes) LMV_FAIL(LMV_ERR_WORKSPACE, "block_bwd: scratch %zu < %zu bytes", scratch_bytes, s.off);
  Side sd{};
  sd.main = (hipStream_t)stream; sd.side = (hipStream_t)side_stream; sd.ws = b.ws_side; sd.ws_bytes = b.ws_side_bytes;
  sd.ws_tail = b.ws_conv; sd.ws_tail_bytes = b.ws_conv_bytes;
  if (side_stream) LMV_TRY(event_pool().take(&sd.fork, &sd.join));
  hipStream_t st = sd.main;
  const int C = D.C, N = D.N, M = D.M;
  lmv_linear_problem p[2];
  lmv_attn_desc ad[2];
  const void* g2[2] = {nullptr, nullptr};
  int rc = LMV_OK;
  auto body = [&]() -> int {
    if (cb) {
      const void* douts[2] = {nullptr, dc_out};
      const float* ds[2] = {nullptr, d->masks[1]};
      LMV_TRY(mlp_bwd(d, D, f, b, 1, douts, ds, nullptr, g2, sd));                    // g2[1] = b.dt2[1]: gradient of c1
      const void* dc1 = b.dt2[1];
      const void* g = dc1;
      if (d->masks[0]) { LMV_TRY(lmv_row_scale(dc1, d->masks[0], b.g2[1], D.rows[1], C, M, D.dtype, st)); g = b.g2[1]; }
      p[0] = prob(g, f.ao[1], d->g_attn_w[2], D.rows[1]); p[0].bias_grad = d->g_attn_b[2];
      LMV_TRY(dw(sd, p, 1, C, C, D.dtype));
      p[0] = prob(g, d->attn_w[2], b.dao[1], D.rows[1]);
      LMV_TRY(lmv_linear_dx(p, 1, C, C, LMV_ACT_NONE, D.dtype, st));
      attn_desc(&ad[0], D, f.pj[1], C, 0, f.pj[0], 2 * C, 0, f.pj[0], 2 * C, C, f.ao[1], f.lse[1], M, N, SDPA_SCALE);
      attn_grads(&ad[0], D, b.dao[1], b.dpj[1], 0, b.dpj[0], 0, b.dpj[0], C);
      LMV_TRY(lmv_attn_bwd(&ad[0], b.ws_main, b.ws_main_bytes, D.dtype, st));
      p[0] = prob(b.dpj[1], f.n1[1], d->g_attn_w[0], D.rows[1]); p[0].bias_grad = d->g_attn_b[0];
      LMV_TRY(dw(sd, p, 1, C, C, D.dtype));
      p[0] = prob(b.dpj[0], f.n1[0], d->g_attn_w[1], D.rows[0]); p[0].bias_grad = d->g_attn_b[1];
      LMV_TRY(dw(sd, p, 1, 2 * C, C, D.dtype));
      p[0] = prob(b.dpj[1], d->attn_w[0], b.dn1[1], D.rows[1]);
      LMV_TRY(lmv_linear_dx(p, 1, C, C, LMV_ACT_NONE, D.dtype, st));
      p[0] = prob(b.dpj[0], d->attn_w[1], b.dn1[0], D.rows[0]);
      LMV_TRY(lmv_linear_dx(p, 1, 2 * C, C, LMV_ACT_NONE, D.dtype, st));
      lmv_ln_segment seg[2] = {};
      seg[0].x = c; seg[0].dy = b.dn1[1]; seg[0].stats = f.st1[1]; seg[0].dres = dc1; seg[0].dx = dc; seg[0].rows = D.rows[1];
      seg[1].x = f.xp; seg[1].dy = b.dn1[0]; seg[1].stats = f.st1[0]; seg[1].dres = nullptr; seg[1].dx = b.dxp; seg[1].rows = D.rows[0];
      LMV_TRY(ln_bwd(sd, seg, 2, d->n1_w, d->g_n1_w, d->g_n1_b, D, b.ws_ln[1], b.ws_ln_bytes));
    } else {
      const void* douts[2] = {dx_out, dc_out};
      const float* ds[2] = {d->masks[1], d->masks[3]};
      const float* nds[2] = {d->masks[0], d->masks[2]};
      LMV_TRY(mlp_bwd(d, D, f, b, 0, douts, ds, nds, g2, sd));
      const bool sh = D.kind == LMV_BLOCK_S;
      for (int s2 = 0; s2 < 2; ++s2) { p[s2] = prob(g2[s2], f.ao[s2], sh ? d->g_attn_w[1] : d->g_attn_w[2 + s2], D.rows[s2]); p[s2].bias_grad = sh ? d->g_attn_b[1] : d->g_attn_b[2 + s2]; }
      LMV_TRY(dw(sd, p, 2, C, C, D.dtype));
      if (sh && d->attn_wt[1] && D.dtype == LMV_BF16) {      // dX of proj / qkv on the transposed weight copies (forward-form GEMM, csrc/wngemm.hip)
        for (int s2 = 0; s2 < 2; ++s2) p[s2] = prob(g2[s2], d->attn_wt[1], b.dao[s2], D.rows[s2]);
        LMV_TRY(lmv_linear_fwd(p, 2, C, C, LMV_ACT_NONE, D.dtype, st));
      } else {
        for (int s2 = 0; s2 < 2; ++s2) p[s2] = prob(g2[s2], sh ? d->attn_w[1] : d->attn_w[2 + s2], b.dao[s2], D.rows[s2]);
        LMV_TRY(lmv_linear_dx(p, 2, C, C, LMV_ACT_NONE, D.dtype, st));
      }
      if (sh) {
        for (int s2 = 0; s2 < 2; ++s2) {
          const int L = s2 == 0 ? N : M;
          attn_desc(&ad[s2], D, f.pj[s2], 3 * C, 0, f.pj[s2], 3 * C, C, f.pj[s2], 3 * C, 2 * C, f.ao[s2], f.lse[s2], L, L, SDPA_SCALE);
          attn_grads(&ad[s2], D, b.dao[s2], b.dpj[s2], 0, b.dpj[s2], C, b.dpj[s2], 2 * C);
        }
        LMV_TRY(lmv_attn_bwd_pair(ad, b.ws_main, b.ws_main_bytes, D.dtype, st));
      } else {
        float sx, sc;
        dca_scales(D, &sx, &sc);
        attn_desc(&ad[0], D, f.pj[0], 3 * C, 0, f.pj[1], 3 * C, C, f.pj[1], 3 * C, 2 * C, f.ao[0], f.lse[0], N, M, sx);
        attn_grads(&ad[0], D, b.dao[0], b.dpj[0], 0, b.dpj[1], C, b.dpj[1], 2 * C);
        attn_desc(&ad[1], D, f.pj[1], 3 * C, 0, f.pj[0], 3 * C, C, f.pj[0], 3 * C, 2 * C, f.ao[1], f.lse[1], M, N, sc);
        attn_grads(&ad[1], D, b.dao[1], b.dpj[1], 0, b.dpj[0], C, b.dpj[0], 2 * C);
        LMV_TRY(lmv_attn_bwd(&ad[0], b.ws_main, b.ws_main_bytes, D.dtype, st));
        LMV_TRY(lmv_attn_bwd(&ad[1], b.ws_main, b.ws_main_bytes, D.dtype, st));
      }
      for (int s2 = 0; s2 < 2; ++s2) { p[s2] = prob(b.dpj[s2], f.n1[s2], sh ? d->g_attn_w[0] : d->g_attn_w[s2], D.rows[s2]); p[s2].bias_grad = sh ? d->g_attn_b[0] : d->g_attn_b[s2]; }
      LMV_TRY(dw(sd, p, 2, 3 * C, C, D.dtype));
      const bool fuse1 = sh && dx_ln_fused_ok(D, d->attn_wt[0], 3 * C);      // dX of qkv + LayerNorm-1 backward in one kernel
      if (fuse1) {
      } else if (sh && d->attn_wt[0] && D.dtype == LMV_BF16) {
        for (int s2 = 0; s2 < 2; ++s2) p[s2] = prob(b.dpj[s2], d->attn_wt[0], b.dn1[s2], D.rows[s2]);
        LMV_TRY(lmv_linear_fwd(p, 2, C, 3 * C, LMV_ACT_NONE, D.dtype, st));
      } else {
        for (int s2 = 0; s2 < 2; ++s2) p[s2] = prob(b.dpj[s2], sh ? d->attn_w[0] : d->attn_w[s2], b.dn1[s2], D.rows[s2]);
        LMV_TRY(lmv_linear_dx(p, 2, 3 * C, C, LMV_ACT_NONE, D.dtype, st));
      }
      lmv_ln_segment seg[2] = {};
      seg[0].x = f.xp; seg[0].dy = b.dn1[0]; seg[0].stats = f.st1[0]; seg[0].dres = b.dt2[0]; seg[0].dx = b.dxp; seg[0].rows = D.rows[0];
      seg[1].x = c; seg[1].dy = b.dn1[1]; seg[1].stats = f.st1[1]; seg[1].dres = b.dt2[1]; seg[1].dx = dc; seg[1].rows = D.rows[1];
      if (fuse1) {
        for (int s2 = 0; s2 < 2; ++s2) p[s2] = prob(b.dpj[s2], d->attn_wt[0], b.dn1[s2], D.rows[s2]);
        LMV_TRY(dx_ln_bwd(sd, p, seg, 2, 3 * C, d->n1_w, d->g_n1_w, d->g_n1_b, D, b.ws_ln[1], b.ws_ln_bytes));
      } else {
        LMV_TRY(ln_bwd(sd, seg, 2, d->n1_w, d->g_n1_w, d->g_n1_b, D, b.ws_ln[1], b.ws_ln_bytes));
      }
    }
    {
      void* ws = sd.ws_tail;
      if (sd.nsegs + 1 > LMV_REDUCE_MAX_SEGS) LMV_FAIL(LMV_ERR_WORKSPACE, "block_bwd: too many deferred reductions");
      int rows = 0;
      LMV_TRY(lmv_dwconv3x3_bwd_weight_partial(b.dxp, x, D.B, D.H, D.W, C, ws, sd.ws_tail_bytes, &rows, D.dtype, sd.begin()));
      lmv_reduce_seg& sg = sd.segs[sd.nsegs++];
      sg = lmv_reduce_seg{};
      sg.ws = (const float*)ws; sg.out_w = d->g_pos_w; sg.out_b = d->g_pos_b; sg.slab_stride = 10 * C; sg.nw = C; sg.nslabs = rows; sg.nb = C;
      sg.kind = LMV_REDUCE_ROWS; sg.mode = 1;
    }
    LMV_TRY(flush_reduces(sd));
    return lmv_dwconv3x3_residual_bwd_data(b.dxp, d->pos_w, dx, D.B, D.H, D.W, C, D.dtype, st);
  };
  rc = body();
  if (rc || !(d->flags & LMV_BLOCK_NO_JOIN)) sd.finish();      // on errors always: never leave the main stream un-joined
  if (side_stream) event_pool().give(sd.fork, sd.join);
  return rc;
}
