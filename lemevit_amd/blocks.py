"""Forward / backward schedules of the three LeMeBlock flavours over the HIP kernels.

Everything here is token-major: x is [B, N, C] (N = H*W image tokens), c is [B, M, C] (meta tokens).
The reference bounces NCHW <-> NLC twice per block (models/lemevit.py:548,579) and re-packs qkv
(:201,290,292,481); here there is no layout change inside a stage and the attention kernels read the
packed projections in place.

  "S" block (:615-650)  x,c share norm1/attn/norm2/mlp weights  -> every Linear is ONE dual-problem launch
  "D" block (:542-582)  x,c share norm1/norm2/mlp; qkv1/qkv2 and proj_x/proj_c differ (dual launch, two weights)
  "C" block (:584-613)  only c is updated; x is returned untouched (:610)

``block_forward`` returns the saved tensors the hand-written ``block_backward`` needs; autograd sees one
node per block (lemevit_amd/model.py::_BlockFn).
"""
from __future__ import annotations

import collections
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .ops import ACT_GELU, ACT_GELU_GRAD, Prob

Tensor = torch.Tensor
BLOCK_LN_EPS = 1e-6   # models/lemevit.py:513,525

# parameter order per block type (names relative to the block; the reference's state_dict keys)
PARAM_NAMES = {
    "S": ["pos_embed.weight", "pos_embed.bias", "norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias",
          "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias"],
    "D": ["pos_embed.weight", "pos_embed.bias", "norm1.weight", "norm1.bias", "attn.qkv1.weight", "attn.qkv1.bias",
          "attn.qkv2.weight", "attn.qkv2.bias", "attn.proj_x.weight", "attn.proj_x.bias", "attn.proj_c.weight", "attn.proj_c.bias",
          "norm2.weight", "norm2.bias", "mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias"],
    "D2": ["pos_embed.weight", "pos_embed.bias", "norm1.weight", "norm1.bias", "attn.qv1.weight", "attn.qv1.bias",
           "attn.kv2.weight", "attn.kv2.bias", "attn.proj_x.weight", "attn.proj_x.bias", "attn.proj_c.weight", "attn.proj_c.bias",
           "norm2.weight", "norm2.bias", "mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias"],
    "C": ["pos_embed.weight", "pos_embed.bias", "norm1.weight", "norm1.bias", "attn.q.weight", "attn.q.bias", "attn.kv.weight", "attn.kv.bias",
          "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias"],
}
# "Sx": the S block of the dense-prediction backbones -- attention and MLP on the image tokens only, the meta tokens pass
# through untouched (object_detection/mmdet/models/backbones/lemevit.py:615-643); same parameters as "S"
PARAM_NAMES["Sx"] = PARAM_NAMES["S"]


# Weight-gradient GEMMs are off the critical path of the backward pass (nothing inside the block consumes dW), so they
# are launched on a SIDE stream next to the dX chain: the ramp-up and tail of every launch, where part of the chip is
# idle, is filled by the other stream's workgroups.  The fork / join are stream waits (capturable into a hipGraph);
# operands of in-flight side launches are kept referenced until the join so the caching allocator cannot recycle them.
_SIDE = os.environ.get("LMV_SIDE_STREAM", "1") != "0"
_side_streams: Dict[int, tuple] = {}          # device index -> (side stream, its raw handle, fork event, join event)
_inflight: List[object] = []


def _fork(dev) -> Optional[int]:
    """Raw handle of the side stream, made to wait for everything enqueued on the current stream so far; None = launch in line."""
    # (measured: -3 % step time in eager mode; inside a hipGraph capture the branch is serialised by the runtime and the
    #  extra dependencies cost 1 %, so captured steps launch the weight gradients in line)
    if not _SIDE or torch.cuda.is_current_stream_capturing():
        return None
    ent = _side_streams.get(dev.index)
    if ent is None:
        side = torch.cuda.Stream(device=dev)
        ent = _side_streams[dev.index] = (side, side.cuda_stream, torch.cuda.Event(), torch.cuda.Event())
    side, raw, fork, _ = ent
    fork.record()                                   # on the current (main) stream: the operands are ready
    side.wait_event(fork)
    return raw


def side_stream_handle(dev) -> Optional[int]:
    """Raw handle of the weight-gradient side stream for the native block schedule (lmv_block_bwd forks / joins it itself); None = in line."""
    if not _SIDE or torch.cuda.is_current_stream_capturing():
        return None
    ent = _side_streams.get(dev.index)
    if ent is None:
        side = torch.cuda.Stream(device=dev)
        ent = _side_streams[dev.index] = (side, side.cuda_stream, torch.cuda.Event(), torch.cuda.Event())
    return ent[1]


_aux_streams: Dict[int, list] = {}


def aux_streams(dev, n: int) -> list:
    """The first n of the device's auxiliary streams (created once, in this order): [0] is the weight-gradient side stream of the backward pass,
    which is idle during a forward pass.  Everything that forks concurrent work (model.image_ranges, graph.split_forward) takes its streams from
    here, so that a process holds main + 3 streams -- the ROCm runtime maps streams onto 4 hardware queues by default, and a stream that lands
    on the main stream's queue serialises behind it (train step with 4 forward ranges on private streams: +0.9 ms)."""
    lst = _aux_streams.setdefault(dev.index, [])
    if not lst:
        ent = _side_streams.get(dev.index)
        if ent is None:
            side = torch.cuda.Stream(device=dev)
            ent = _side_streams[dev.index] = (side, side.cuda_stream, torch.cuda.Event(), torch.cuda.Event())
        lst.append(ent[0])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=dev))
    return lst[:n]


# The split-K reductions of a block's weight gradients are deferred (ops.DwBatch) and summed by ONE launch at the end of the block's
# backward pass instead of one ~7 us launch behind every GEMM (LMV_DW_BATCH=0: the per-GEMM path, for A/B runs).
_DW_BATCH = False      # (a module attribute for A/B runs, no environment switch) measured: 0.5 ms SLOWER per step (the slabs of a whole block leave the MALL before they are read back)
_batches: Dict[tuple, "ops.DwBatch"] = {}


def _dw(probs, N: int, K: int) -> None:
    dev = probs[0].a.device
    raw = _fork(dev)
    batch = None
    if _DW_BATCH:
        batch = _batches.get((dev.index, raw))
        if batch is None:
            batch = _batches[(dev.index, raw)] = ops.DwBatch()
    ops.linear_dw(probs, N, K, stream=raw, batch=batch)          # launched on the side stream by handle: no current-stream switch
    if raw is not None:
        _inflight.append((dev.index, probs))
    elif batch is not None:
        batch.keep.append(probs)


def _dwconv_w(dy: Tensor, x: Tensor, dweight: Tensor, dbias: Tensor, H: int, W: int) -> None:
    raw = _fork(dy.device)
    ops.dwconv_bwd_weight(dy, x, dweight, dbias, H, W, stream=raw)
    if raw is not None:
        _inflight.append((dy.device.index, (dy, x)))


# The join of a block's side-stream launches is DEFERRED: the main stream does not wait at the end of the block that issued them (its
# last weight-gradient GEMM and the dwconv weight gradient were requested moments before: the main stream idled ~10-20 us per block
# there) but at the end of the NEXT block's backward pass, when they have long finished; whatever they read stays referenced until
# then.  The last blocks of a backward pass are joined by an autograd final callback, i.e. before .backward() returns to the caller.
_DEFER = int(os.environ.get("LMV_JOIN_DEFER", "1"))       # blocks a join may trail behind (0: join at the end of every block)
_pending: "Dict[int, collections.deque]" = {}              # device index -> deque of (event recorded on that device's side stream, references)
_event_pool: "Dict[int, List[torch.cuda.Event]]" = {}      # device index -> reusable events
_cb_token = None                                           # identity of the backward pass whose final callback is queued (see defer_join)


def _wait_pending(keep: int, dev_index: Optional[int] = None) -> None:
    """Make the CURRENT stream of each device wait for its deferred side-stream positions (all but the `keep` newest)."""
    for di in ([dev_index] if dev_index is not None else list(_pending)):
        q = _pending.get(di)
        while q and len(q) > keep:
            ev, _refs = q.popleft()
            torch.cuda.current_stream(torch.device("cuda", di)).wait_event(ev)
            _event_pool.setdefault(di, []).append(ev)


def drain_deferred() -> None:
    """Join everything that is still deferred and forget the queued-callback marker.  Called at the start of every training-mode forward
    pass (model.new_training_pass), by FlatAdamW.step() and by FlatGradSync.finish(): when a backward pass RAISES (out of memory,
    anomaly mode), autograd drops its queued final callbacks, so relying on `_final_join` alone would leave the last blocks un-joined --
    and a sticky "callback queued" flag would keep every later backward pass from queueing its own."""
    global _cb_token
    _cb_token = None
    _wait_pending(0)


def _final_join() -> None:
    global _cb_token
    _cb_token = None
    _wait_pending(0)


def defer_join(dev_index: int, refs) -> None:
    """Record the side stream's position; the main stream waits for it `_DEFER` blocks later (or when the backward pass ends)."""
    global _cb_token
    side = _side_streams[dev_index][0]
    pool = _event_pool.setdefault(dev_index, [])
    ev = pool.pop() if pool else torch.cuda.Event()
    ev.record(side)
    _pending.setdefault(dev_index, collections.deque()).append((ev, refs))
    keep = _DEFER
    if keep > 0:
        # one final callback per backward pass: the marker is the pass's graph task (a new pass -- also one that follows a pass which
        # raised before its callbacks ran -- has a different one), not a process-wide flag
        try:
            token = torch._C._current_graph_task_id()
        except Exception:
            token = -1
        if token == -1:                   # not inside a backward pass (a schedule driven by hand): join now
            keep = 0
        elif _cb_token != token:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_final_join)
                _cb_token = token
            except RuntimeError:
                keep = 0
    _wait_pending(keep, dev_index)


def _join() -> None:
    for (di, raw), batch in _batches.items():
        if batch.segs:
            batch.flush(stream=raw)                 # one reduce launch for every weight gradient of the block, behind its GEMMs
    if _inflight:
        refs = list(_inflight)
        _inflight.clear()
        defer_join(refs[-1][0], refs)


# The meta-token self-attention of an S block (16 tokens: B * h tiny workgroups, ~13 us of mostly launch ramp and tail per
# direction) is independent of the image-token attention next to it: it is launched on a SECOND side stream by raw handle and
# joined before the projection that consumes both, so it runs inside the image-token kernel's shadow.
_META_SIDE = False      # (a module attribute for A/B runs, no environment switch) measured: 0.3 ms SLOWER per step (4 stream-wait API calls per block for ~20 us of hidden kernels); kept as an A/B switch
_meta_streams: Dict[int, tuple] = {}


def _meta_fork(dev) -> Optional[int]:
    if not _META_SIDE or torch.cuda.is_current_stream_capturing():
        return None
    ent = _meta_streams.get(dev.index)
    if ent is None:
        side = torch.cuda.Stream(device=dev)
        ent = _meta_streams[dev.index] = (side, side.cuda_stream, torch.cuda.Event(), torch.cuda.Event())
    side, raw, fork, _ = ent
    fork.record()
    side.wait_event(fork)
    return raw


def _meta_join(dev) -> None:
    side, _, _, join = _meta_streams[dev.index]
    join.record(side)
    torch.cuda.current_stream(dev).wait_event(join)


def _empty(rows_like: Tensor, cols: int) -> Tensor:
    return torch.empty(rows_like.shape[:-1] + (cols,), device=rows_like.device, dtype=rows_like.dtype)


def _rps(t: Tensor) -> int:
    return t.shape[1]


# ------------------------------------------------------------------------------------------------
# MLP half of a block:  t <- t + ds * fc2(GELU(fc1(LN2(t))))   for every stream t in `ts`
# ------------------------------------------------------------------------------------------------
def _lib_config(key: str) -> int:
    from . import _lib
    return _lib.config_get(key)


def _mlp_fwd(P: Dict[str, Tensor], ts: Sequence[Tensor], ds: Sequence[Optional[Tensor]], save: bool, fc1_fold=None, pre_ln=None):
    C = ts[0].shape[-1]
    Hd = P["mlp.0.weight"].shape[0]
    if pre_ln is None and fc1_fold is not None and not save and ops.mlp_fused_supported(C, Hd, ts[0].dtype):
        # inference: LN2 -> fc1 -> GELU -> fc2 -> + residual in ONE kernel, the hidden activations never leave the chip
        return ops.mlp_fused_fwd(ts, fc1_fold, P["mlp.3.weight"], P["mlp.3.bias"], BLOCK_LN_EPS, ds), None
    if pre_ln is not None:                       # norm2 came out of the attention projection's launch (_attn_S_fwd, ops.linear_res_ln_fwd)
        xn, st = pre_ln
    else:
        xn, st = ops.layernorm_fwd_multi(ts, P["norm2.weight"], P["norm2.bias"], BLOCK_LN_EPS, want_stats=save)
    h = [_empty(t, Hd) for t in ts]
    u = [_empty(t, Hd) if save else None for t in ts]
    ops.linear_fwd([Prob(a, P["mlp.0.weight"], o, bias=P["mlp.0.bias"], out_pre=pre) for a, o, pre in zip(xn, h, u)], Hd, C, ACT_GELU)
    out = [torch.empty_like(t) for t in ts]
    ops.linear_fwd([Prob(a, P["mlp.3.weight"], o, bias=P["mlp.3.bias"], res=t, row_scale=s, rps=_rps(t)) for a, o, t, s in zip(h, out, ts, ds)], C, Hd)
    saved = (list(ts), list(st), list(xn), u, h) if save else None
    return out, saved


def _mlp_bwd(P, G, saved, douts: Sequence[Tensor], ds: Sequence[Optional[Tensor]], next_ds: Optional[Sequence[Optional[Tensor]]] = None):
    """next_ds: DropPath vectors of the attention half that is differentiated next; the closing LayerNorm-backward launch then
    also writes the gradient pre-scaled by them (returned as the second list) -- no separate row-scale pass."""
    ts, st, xn, u, h = saved
    C = ts[0].shape[-1]
    Hd = P["mlp.0.weight"].shape[0]
    g = ops.row_scale_multi(douts, ds)
    _dw([Prob(gi, hi, G["mlp.3.weight"], bias_grad=G["mlp.3.bias"]) for gi, hi in zip(g, h)], C, Hd)
    du = [torch.empty_like(ui) for ui in u]
    ops.linear_dx([Prob(gi, P["mlp.3.weight"], o, aux=ui) for gi, o, ui in zip(g, du, u)], C, Hd, ACT_GELU_GRAD)
    _dw([Prob(dui, xi, G["mlp.0.weight"], bias_grad=G["mlp.0.bias"]) for dui, xi in zip(du, xn)], Hd, C)
    dxn = [torch.empty_like(t) for t in ts]
    ops.linear_dx([Prob(dui, P["mlp.0.weight"], o) for dui, o in zip(du, dxn)], Hd, C)
    return ops.layernorm_bwd_multi(dxn, ts, st, P["norm2.weight"], G["norm2.weight"], G["norm2.bias"], douts, next_scales=next_ds)


# ------------------------------------------------------------------------------------------------
# attention halves
# ------------------------------------------------------------------------------------------------
def _attn_S_fwd(P, ts, ds, save, want_ln2=False):
    """t <- t + ds * proj(SA(qkv(LN1(t)))) for x and c with the SAME weights (models/lemevit.py:632,634).
    want_ln2: where the library fuses the projection with the norm2 that follows it (ops.res_ln_fused), run that launch and return
    norm2's outputs as a third element: (xn2, st2) or None."""
    C = ts[0].shape[-1]
    xn, st = ops.layernorm_fwd_multi(ts, P["norm1.weight"], P["norm1.bias"], BLOCK_LN_EPS, want_stats=save)
    qkv = [_empty(t, 3 * C) for t in ts]
    ops.linear_fwd([Prob(a, P["attn.qkv.weight"], o, bias=P["attn.qkv.bias"]) for a, o in zip(xn, qkv)], 3 * C, C)
    if len(qkv) == 2 and not _META_SIDE:
        ao, lse = ops.attn_fwd_pair(qkv, C, ops.SDPA_SCALE, want_lse=save)      # image tokens + meta tokens: one launch
    elif len(qkv) == 2:
        raw = _meta_fork(qkv[1].device)
        ao_c, lse_c = ops.attn_fwd((qkv[1], 0), (qkv[1], C), (qkv[1], 2 * C), C, ops.SDPA_SCALE, want_lse=save, stream=raw)
        ao_x, lse_x = ops.attn_fwd((qkv[0], 0), (qkv[0], C), (qkv[0], 2 * C), C, ops.SDPA_SCALE, want_lse=save)
        if raw is not None:
            _meta_join(qkv[1].device)
        ao, lse = (ao_x, ao_c), (lse_x, lse_c)
    else:
        ao, lse = zip(*[ops.attn_fwd((q, 0), (q, C), (q, 2 * C), C, ops.SDPA_SCALE, want_lse=save) for q in qkv])
    out = [torch.empty_like(t) for t in ts]
    probs = [Prob(a, P["attn.proj.weight"], o, bias=P["attn.proj.bias"], res=t, row_scale=s, rps=_rps(t)) for a, o, t, s in zip(ao, out, ts, ds)]
    saved = (list(ts), list(st), list(xn), qkv, list(ao), list(lse)) if save else None
    if want_ln2:
        ln2 = None
        if ops.res_ln_fused(C, C, sum(p.rows for p in probs), ts[0].dtype):
            ln2 = ops.linear_res_ln_fwd(probs, C, C, P["norm2.weight"], P["norm2.bias"], BLOCK_LN_EPS, want_stats=save)
        else:
            ops.linear_fwd(probs, C, C)
        return out, saved, ln2
    ops.linear_fwd(probs, C, C)
    return out, saved


def _attn_S_bwd(P, G, saved, douts, ds, g=None):
    ts, st, xn, qkv, ao, lse = saved
    C = ts[0].shape[-1]
    g = ops.row_scale_multi(douts, ds) if g is None else g
    _dw([Prob(gi, ai, G["attn.proj.weight"], bias_grad=G["attn.proj.bias"]) for gi, ai in zip(g, ao)], C, C)
    dao = [torch.empty_like(t) for t in ts]
    ops.linear_dx([Prob(gi, P["attn.proj.weight"], o) for gi, o in zip(g, dao)], C, C)
    dqkv = [torch.empty_like(q) for q in qkv]
    if len(qkv) == 2 and not _META_SIDE:
        ops.attn_bwd_pair(qkv, ao, lse, dao, dqkv, C, ops.SDPA_SCALE)
    else:
        raw = _meta_fork(qkv[1].device) if len(qkv) == 2 else None
        for i in reversed(range(len(qkv))):              # meta tokens first (side stream), image tokens on the main stream
            q, a, l, da, dq = qkv[i], ao[i], lse[i], dao[i], dqkv[i]
            ops.attn_bwd((q, 0), (q, C), (q, 2 * C), a, l, da, (dq, 0), (dq, C), (dq, 2 * C), C, ops.SDPA_SCALE, stream=raw if i == 1 else None)
        if raw is not None:
            _meta_join(qkv[1].device)
    _dw([Prob(dq, xi, G["attn.qkv.weight"], bias_grad=G["attn.qkv.bias"]) for dq, xi in zip(dqkv, xn)], 3 * C, C)
    dxn = [torch.empty_like(t) for t in ts]
    ops.linear_dx([Prob(dq, P["attn.qkv.weight"], o) for dq, o in zip(dqkv, dxn)], 3 * C, C)
    return ops.layernorm_bwd_multi(dxn, ts, st, P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], douts)


def _attn_D_fwd(P, ts, ds, save):
    """Dual cross attention (models/lemevit.py:252-256,288-302): ts = [x, c]."""
    x, c = ts
    C, N, M = x.shape[-1], x.shape[1], c.shape[1]
    sx, sc = ops.dca_scales(N, M, C)
    q1, q2 = _empty(x, 3 * C), _empty(c, 3 * C)
    if ops.ln_exact_fused(3 * C, C, x.dtype):        # norm1 and both projections in one launch (csrc/rswgemm.hip, C = 96)
        xn, st = ops.ln_linear_exact_fwd([Prob(x, P["attn.qkv1.weight"], q1, bias=P["attn.qkv1.bias"]), Prob(c, P["attn.qkv2.weight"], q2, bias=P["attn.qkv2.bias"])],
                                         3 * C, C, P["norm1.weight"], P["norm1.bias"], BLOCK_LN_EPS, want_stats=save, want_ln=save)
    else:
        xn, st = ops.layernorm_fwd_multi(ts, P["norm1.weight"], P["norm1.bias"], BLOCK_LN_EPS, want_stats=save)
        ops.linear_fwd([Prob(xn[0], P["attn.qkv1.weight"], q1, bias=P["attn.qkv1.bias"]),
                        Prob(xn[1], P["attn.qkv2.weight"], q2, bias=P["attn.qkv2.bias"])], 3 * C, C)
    aox, lsex = ops.attn_fwd((q1, 0), (q2, C), (q2, 2 * C), C, sx, want_lse=save)     # image -> meta   (:297)
    aoc, lsec = ops.attn_fwd((q2, 0), (q1, C), (q1, 2 * C), C, sc, want_lse=save)     # meta  -> image  (:300)
    ox, oc = torch.empty_like(x), torch.empty_like(c)
    ops.linear_fwd([Prob(aox, P["attn.proj_x.weight"], ox, bias=P["attn.proj_x.bias"], res=x, row_scale=ds[0], rps=N),
                    Prob(aoc, P["attn.proj_c.weight"], oc, bias=P["attn.proj_c.bias"], res=c, row_scale=ds[1], rps=M)], C, C)
    return [ox, oc], ((list(ts), list(st), list(xn), q1, q2, aox, aoc, lsex, lsec) if save else None)


def _attn_D_bwd(P, G, saved, douts, ds, g=None):
    ts, st, xn, q1, q2, aox, aoc, lsex, lsec = saved
    x, c = ts
    C, N, M = x.shape[-1], x.shape[1], c.shape[1]
    sx, sc = ops.dca_scales(N, M, C)
    g = ops.row_scale_multi(douts, ds) if g is None else g
    _dw([Prob(g[0], aox, G["attn.proj_x.weight"], bias_grad=G["attn.proj_x.bias"]),
                   Prob(g[1], aoc, G["attn.proj_c.weight"], bias_grad=G["attn.proj_c.bias"])], C, C)
    daox, daoc = torch.empty_like(x), torch.empty_like(c)
    ops.linear_dx([Prob(g[0], P["attn.proj_x.weight"], daox), Prob(g[1], P["attn.proj_c.weight"], daoc)], C, C)
    dq1, dq2 = torch.empty_like(q1), torch.empty_like(q2)
    ops.attn_bwd((q1, 0), (q2, C), (q2, 2 * C), aox, lsex, daox, (dq1, 0), (dq2, C), (dq2, 2 * C), C, sx)
    ops.attn_bwd((q2, 0), (q1, C), (q1, 2 * C), aoc, lsec, daoc, (dq2, 0), (dq1, C), (dq1, 2 * C), C, sc)
    _dw([Prob(dq1, xn[0], G["attn.qkv1.weight"], bias_grad=G["attn.qkv1.bias"]),
                   Prob(dq2, xn[1], G["attn.qkv2.weight"], bias_grad=G["attn.qkv2.bias"])], 3 * C, C)
    dxn = [torch.empty_like(x), torch.empty_like(c)]
    ops.linear_dx([Prob(dq1, P["attn.qkv1.weight"], dxn[0]), Prob(dq2, P["attn.qkv2.weight"], dxn[1])], 3 * C, C)
    return ops.layernorm_bwd_multi(dxn, ts, st, P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], douts)


def _attn_D2_fwd(P, ts, ds, save):
    """DualCrossAttention_v2 ("D2", models/lemevit.py:357-361,393-407): q, v1 from the image tokens, k, v2 from the meta tokens;
    the SAME q / k serve both directions (x' = sdpa(q, k, v2), c' = sdpa(k, q, v1))."""
    x, c = ts
    C, N, M = x.shape[-1], x.shape[1], c.shape[1]
    sx, sc = ops.dca_scales(N, M, C)
    xn, st = ops.layernorm_fwd_multi(ts, P["norm1.weight"], P["norm1.bias"], BLOCK_LN_EPS, want_stats=save)
    qv1, kv2 = _empty(x, 2 * C), _empty(c, 2 * C)
    ops.linear_fwd([Prob(xn[0], P["attn.qv1.weight"], qv1, bias=P["attn.qv1.bias"]),
                    Prob(xn[1], P["attn.kv2.weight"], kv2, bias=P["attn.kv2.bias"])], 2 * C, C)
    aox, lsex = ops.attn_fwd((qv1, 0), (kv2, 0), (kv2, C), C, sx, want_lse=save)     # :402
    aoc, lsec = ops.attn_fwd((kv2, 0), (qv1, 0), (qv1, C), C, sc, want_lse=save)     # :405
    ox, oc = torch.empty_like(x), torch.empty_like(c)
    ops.linear_fwd([Prob(aox, P["attn.proj_x.weight"], ox, bias=P["attn.proj_x.bias"], res=x, row_scale=ds[0], rps=N),
                    Prob(aoc, P["attn.proj_c.weight"], oc, bias=P["attn.proj_c.bias"], res=c, row_scale=ds[1], rps=M)], C, C)
    return [ox, oc], ((list(ts), list(st), list(xn), qv1, kv2, aox, aoc, lsex, lsec) if save else None)


def _attn_D2_bwd(P, G, saved, douts, ds, g=None):
    ts, st, xn, qv1, kv2, aox, aoc, lsex, lsec = saved
    x, c = ts
    C, N, M = x.shape[-1], x.shape[1], c.shape[1]
    sx, sc = ops.dca_scales(N, M, C)
    g = ops.row_scale_multi(douts, ds) if g is None else g
    _dw([Prob(g[0], aox, G["attn.proj_x.weight"], bias_grad=G["attn.proj_x.bias"]),
                   Prob(g[1], aoc, G["attn.proj_c.weight"], bias_grad=G["attn.proj_c.bias"])], C, C)
    daox, daoc = torch.empty_like(x), torch.empty_like(c)
    ops.linear_dx([Prob(g[0], P["attn.proj_x.weight"], daox), Prob(g[1], P["attn.proj_c.weight"], daoc)], C, C)
    dqv1, dkv2 = torch.empty_like(qv1), torch.empty_like(kv2)
    ops.attn_bwd((qv1, 0), (kv2, 0), (kv2, C), aox, lsex, daox, (dqv1, 0), (dkv2, 0), (dkv2, C), C, sx)
    # second direction: k is the query, q the key -> their gradients ADD to the ones above (small torch adds, "D2" only)
    # (gradient tensors carry the strides of the operand they belong to: scratch copies of the packed layouts)
    t_kv2, t_qv1 = torch.empty_like(kv2), torch.empty_like(qv1)
    ops.attn_bwd((kv2, 0), (qv1, 0), (qv1, C), aoc, lsec, daoc, (t_kv2, 0), (t_qv1, 0), (dqv1, C), C, sc)
    dqv1[..., :C] += t_qv1[..., :C]
    dkv2[..., :C] += t_kv2[..., :C]
    _dw([Prob(dqv1, xn[0], G["attn.qv1.weight"], bias_grad=G["attn.qv1.bias"]),
                   Prob(dkv2, xn[1], G["attn.kv2.weight"], bias_grad=G["attn.kv2.bias"])], 2 * C, C)
    dxn = [torch.empty_like(x), torch.empty_like(c)]
    ops.linear_dx([Prob(dqv1, P["attn.qv1.weight"], dxn[0]), Prob(dkv2, P["attn.kv2.weight"], dxn[1])], 2 * C, C)
    return ops.layernorm_bwd_multi(dxn, ts, st, P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], douts)


def _attn_C_fwd(P, xp, c, ds, save):
    """c <- c + ds * proj(CA(q(LN1(c)), kv(LN1(xp))))  (models/lemevit.py:477-486,600)."""
    C, N, M = c.shape[-1], xp.shape[1], c.shape[1]
    kv, q = _empty(xp, 2 * C), _empty(c, C)
    if ops.ln_exact_fused(2 * C, C, xp.dtype) and ops.ln_exact_fused(C, C, xp.dtype):      # norm1 inside the two projection launches (C = 96)
        (xn,), (stx,) = ops.ln_linear_exact_fwd([Prob(xp, P["attn.kv.weight"], kv, bias=P["attn.kv.bias"])], 2 * C, C, P["norm1.weight"], P["norm1.bias"], BLOCK_LN_EPS, want_stats=save, want_ln=save)
        (cn,), (stc,) = ops.ln_linear_exact_fwd([Prob(c, P["attn.q.weight"], q, bias=P["attn.q.bias"])], C, C, P["norm1.weight"], P["norm1.bias"], BLOCK_LN_EPS, want_stats=save, want_ln=save)
    else:
        (xn, cn), (stx, stc) = ops.layernorm_fwd_multi([xp, c], P["norm1.weight"], P["norm1.bias"], BLOCK_LN_EPS, want_stats=save)
        ops.linear_fwd([Prob(xn, P["attn.kv.weight"], kv, bias=P["attn.kv.bias"])], 2 * C, C)
        ops.linear_fwd([Prob(cn, P["attn.q.weight"], q, bias=P["attn.q.bias"])], C, C)
    ao, lse = ops.attn_fwd((q, 0), (kv, 0), (kv, C), C, ops.SDPA_SCALE, want_lse=save)
    oc = torch.empty_like(c)
    ops.linear_fwd([Prob(ao, P["attn.proj.weight"], oc, bias=P["attn.proj.bias"], res=c, row_scale=ds, rps=M)], C, C)
    return oc, ((xp, c, stx, stc, xn, cn, kv, q, ao, lse) if save else None)


def _attn_C_bwd(P, G, saved, dout, ds):
    xp, c, stx, stc, xn, cn, kv, q, ao, lse = saved
    C, M = c.shape[-1], c.shape[1]
    g = dout if ds is None else ops.row_scale(dout, ds, M)
    _dw([Prob(g, ao, G["attn.proj.weight"], bias_grad=G["attn.proj.bias"])], C, C)
    dao = torch.empty_like(c)
    ops.linear_dx([Prob(g, P["attn.proj.weight"], dao)], C, C)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    ops.attn_bwd((q, 0), (kv, 0), (kv, C), ao, lse, dao, (dq, 0), (dkv, 0), (dkv, C), C, ops.SDPA_SCALE)
    _dw([Prob(dq, cn, G["attn.q.weight"], bias_grad=G["attn.q.bias"])], C, C)
    _dw([Prob(dkv, xn, G["attn.kv.weight"], bias_grad=G["attn.kv.bias"])], 2 * C, C)
    dcn, dxn = torch.empty_like(c), torch.empty_like(xp)
    ops.linear_dx([Prob(dq, P["attn.q.weight"], dcn)], C, C)
    ops.linear_dx([Prob(dkv, P["attn.kv.weight"], dxn)], 2 * C, C)
    dc, dxp = ops.layernorm_bwd_multi([dcn, dxn], [c, xp], [stc, stx], P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], [dout, None])
    return dxp, dc


# ------------------------------------------------------------------------------------------------
# whole blocks
# ------------------------------------------------------------------------------------------------
def block_forward(kind: str, x: Tensor, c: Tensor, H: int, W: int, P: Dict[str, Tensor],
                  masks: Sequence[Optional[Tensor]], save: bool, folds=None):
    """LeMeBlock.forward (models/lemevit.py:652-660) on token-major x [B,HW,C], c [B,M,C].
    masks: per-sample DropPath scales in the reference's draw order (D/S: x-attn, x-mlp, c-attn, c-mlp; C: c-attn, c-mlp)."""
    xp = ops.dwconv_residual_fwd(x, P["pos_embed.weight"], P["pos_embed.bias"], H, W)          # :546
    fc1_fold = None if folds is None else folds[1]      # (lemevit_amd.model.block_folds; the attention halves keep LayerNorm + Linear here)
    if kind == "C":
        c1, sa = _attn_C_fwd(P, xp, c, masks[0], save)
        (c2,), sm = _mlp_fwd(P, [c1], [masks[1]], save, fc1_fold)
        return x, c2, ((x, sa, sm) if save else None)                                            # returns the ORIGINAL x (:610)
    if kind == "Sx":
        (x2,), sa = _attn_S_fwd(P, [xp], [masks[0]], save)
        (x3,), sm = _mlp_fwd(P, [x2], [masks[1]], save, fc1_fold)
        return x3, c, ((x, sa, sm) if save else None)
    if kind == "S":
        # (the one-kernel / LayerNorm-folded MLP half of the fused inference schedule has no use for norm2's output: csrc/block.hip::res_ln_ok)
        Hd = P["mlp.0.weight"].shape[0]
        rows = xp.numel() // xp.shape[-1] + c.numel() // c.shape[-1]
        split = xp.shape[-1] == 384 and Hd == 1536 and 16384 <= rows <= 65536 and bool(_lib_config("mlp_split384"))      # csrc/block.hip::mlp_fwd: LayerNorm + rsgemm fc1 + wngemm fc2
        folded = fc1_fold is not None and not save and not split
        r = _attn_S_fwd(P, [xp, c], [masks[0], masks[2]], save, want_ln2=not folded)
        (x2, c1), sa, ln2 = r if not folded else (r[0], r[1], None)
        # split: never the one-kernel MLP, with or without a fused norm2 in front (the native schedule's rule)
        (x3, c2), sm = _mlp_fwd(P, [x2, c1], [masks[1], masks[3]], save, None if split else fc1_fold, pre_ln=ln2)
        return x3, c2, ((x, sa, sm) if save else None)
    fwd = {"D": _attn_D_fwd, "D2": _attn_D2_fwd}[kind]
    (x2, c1), sa = fwd(P, [xp, c], [masks[0], masks[2]], save)
    (x3, c2), sm = _mlp_fwd(P, [x2, c1], [masks[1], masks[3]], save, fc1_fold)
    return x3, c2, ((x, sa, sm) if save else None)


def block_backward(kind: str, saved, dx: Tensor, dc: Tensor, H: int, W: int, P: Dict[str, Tensor], G: Dict[str, Tensor],
                   masks: Sequence[Optional[Tensor]]) -> Tuple[Tensor, Tensor]:
    """Gradients wrt the block inputs; parameter gradients are accumulated (fp32) into G."""
    x0, sa, sm = saved
    if kind == "C":
        (dc1,) = _mlp_bwd(P, G, sm, [dc], [masks[1]])
        dxp, dc0 = _attn_C_bwd(P, G, sa, dc1, masks[0])
        _dwconv_w(dxp, x0, G["pos_embed.weight"], G["pos_embed.bias"], H, W)
        dx0 = ops.dwconv_residual_bwd_data(dxp, P["pos_embed.weight"], H, W)
        _join()
        return (dx0 if dx is None else dx0 + dx), dc0   # the untouched x's pass-through gradient is added by autograd
    if kind == "Sx":
        (dx2,), g_attn = _mlp_bwd(P, G, sm, [dx], [masks[1]], next_ds=[masks[0]])
        (dxp,) = _attn_S_bwd(P, G, sa, [dx2], [masks[0]], g=g_attn)
        _dwconv_w(dxp, x0, G["pos_embed.weight"], G["pos_embed.bias"], H, W)
        dx0 = ops.dwconv_residual_bwd_data(dxp, P["pos_embed.weight"], H, W)
        _join()
        return dx0, dc                                      # the meta tokens' gradient passes through
    (dx2, dc1), g_attn = _mlp_bwd(P, G, sm, [dx, dc], [masks[1], masks[3]], next_ds=[masks[0], masks[2]])
    bwd = {"S": _attn_S_bwd, "D": _attn_D_bwd, "D2": _attn_D2_bwd}[kind]
    dxp, dc0 = bwd(P, G, sa, [dx2, dc1], [masks[0], masks[2]], g=g_attn)
    _dwconv_w(dxp, x0, G["pos_embed.weight"], G["pos_embed.bias"], H, W)
    dx0 = ops.dwconv_residual_bwd_data(dxp, P["pos_embed.weight"], H, W)
    _join()
    return dx0, dc0
