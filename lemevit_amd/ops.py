"""Thin torch-tensor wrappers over the C ABI (include/lemevit_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every op below enqueues hand-written
HIP kernels on ``torch.cuda.current_stream()`` and nothing else.  Tensors must live on a GPU: there is
no CPU path (the CPU restatement lives under ``oracle/`` and is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_GELU_GRAD, ACT_NONE, AttnDesc, LinearProblem, LnSegment, check, lib

Tensor = torch.Tensor
HEAD_DIM = 32


def dtype_code(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.LMV_F32
    if t.dtype == torch.bfloat16:
        return _lib.LMV_BF16
    raise TypeError(f"lemevit_amd: unsupported dtype {t.dtype} (float32 and bfloat16 only)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Raw handle of the current HIP stream of the current device.  (torch.cuda.current_stream().cuda_stream builds a Python
    Stream object per call: ~8 us, i.e. ~6 ms of host time per train step at ~700 launches.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("lemevit_amd: tensors must be on the GPU (no CPU fallback exists)")
    if not t.is_contiguous():
        raise RuntimeError("lemevit_amd: tensors must be contiguous")
    return t.data_ptr()


_ws_cache = {}


def _workspace(nbytes: int, device, stream: Optional[int] = None) -> Tensor:
    """Scratch for the split reductions, one per (device, stream); stream-ordered reuse is safe because every consumer
    of the scratch is enqueued on the same stream before the next producer."""
    key = (device, _stream() if stream is None else stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None and stream is not None:
            # The caching allocator only knows the stream that was current at allocation time.  A launch enqueued by raw
            # handle on another stream may still be using the old scratch: tell the allocator, so the block is not handed
            # out again before that stream has passed this point.
            ws.record_stream(torch.cuda.ExternalStream(stream, device=device))
        ws = torch.empty(max(int(nbytes * 1.25), 1 << 22), device=device, dtype=torch.uint8)
        _ws_cache[key] = ws
    return ws


def _f32(t: Optional[Tensor]) -> Optional[int]:
    if t is not None and t.dtype != torch.float32:
        raise TypeError("lemevit_amd: vector operands (bias, LN affine, scales, gradient accumulators) must be float32")
    return _ptr(t)


# -------------------------------------------------------------------------------------------
# Linear
# -------------------------------------------------------------------------------------------
class Prob:
    """One problem of a (dual) linear launch; see lmv_linear_problem."""
    __slots__ = ("a", "w", "bias", "res", "row_scale", "aux", "out", "out_pre", "bias_grad", "rows", "rps")

    def __init__(self, a, w, out, bias=None, res=None, row_scale=None, aux=None, out_pre=None, bias_grad=None, rps=0):
        self.a, self.w, self.out, self.bias, self.res = a, w, out, bias, res
        self.row_scale, self.aux, self.out_pre, self.bias_grad, self.rps = row_scale, aux, out_pre, bias_grad, rps
        self.rows = a.numel() // a.shape[-1]


def _pack(probs: Sequence[Prob]):
    arr = (LinearProblem * len(probs))()
    for s, p in zip(arr, probs):
        s.a, s.w, s.out = _ptr(p.a), _ptr(p.w), _ptr(p.out)
        s.bias, s.row_scale, s.bias_grad = _f32(p.bias), _f32(p.row_scale), _f32(p.bias_grad)
        s.res, s.aux, s.out_pre = _ptr(p.res), _ptr(p.aux), _ptr(p.out_pre)
        s.rows, s.rows_per_sample = p.rows, p.rps
    return arr


def linear_fwd(probs: Sequence[Prob], N: int, K: int, act: int = ACT_NONE) -> None:
    """out = res + row_scale * act(a @ w^T + bias) for up to two problems sharing (N, K)."""
    check(lib.lmv_linear_fwd(_pack(probs), len(probs), N, K, act, dtype_code(probs[0].a), _stream()), "lmv_linear_fwd")


def linear_dx(probs: Sequence[Prob], N: int, K: int, act: int = ACT_NONE) -> None:
    """out[r,k] = (a[r,:] @ w) (* gelu'(aux)) (+ res); a = dY [rows,N], w = W [N,K]."""
    check(lib.lmv_linear_dx(_pack(probs), len(probs), N, K, act, dtype_code(probs[0].a), _stream()), "lmv_linear_dx")


# -------------------------------------------------------------------------------------------
# Fused entry points: LayerNorm folded into the consuming Linear, the whole MLP half in one kernel
# -------------------------------------------------------------------------------------------
class Folded:
    """(W', colsum, b') of lmv_ln_fold: LN(x) W^T + b = rstd (x W'^T - mean colsum) + b'."""
    __slots__ = ("wf", "colsum", "bf")

    def __init__(self, wf, colsum, bf):
        self.wf, self.colsum, self.bf = wf, colsum, bf


def ln_fold(weight: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, dtype: torch.dtype) -> Folded:
    """Fold LayerNorm(gamma, beta) into the Linear (weight [N, K] fp32 master, bias [N] or None) that consumes it."""
    N, K = weight.shape
    if weight.dtype != torch.float32:
        raise TypeError("lemevit_amd: ln_fold takes the fp32 master weight")
    wf = torch.empty((N, K), device=weight.device, dtype=dtype)
    colsum = torch.empty(N, device=weight.device, dtype=torch.float32)
    bf = torch.empty(N, device=weight.device, dtype=torch.float32)
    check(lib.lmv_ln_fold(_ptr(weight), _f32(bias), _f32(gamma), _f32(beta), N, K, _ptr(wf), _ptr(colsum), _ptr(bf), dtype_code(wf), _stream()), "lmv_ln_fold")
    return Folded(wf, colsum, bf)


def ln_linear_fwd(probs: Sequence[Prob], N: int, K: int, eps: float, act: int = ACT_NONE) -> None:
    """out = res + row_scale * act(LN(a) @ W^T + b) from folded operands: Prob(a=x, w=F.wf, out, bias=F.bf, aux=F.colsum)."""
    arr = _pack(probs)
    for s, p in zip(arr, probs):
        s.aux = _f32(p.aux)
    check(lib.lmv_ln_linear_fwd(arr, len(probs), N, K, eps, act, dtype_code(probs[0].a), _stream()), "lmv_ln_linear_fwd")


def mlp_fused_supported(C_: int, hidden: int, dtype: torch.dtype) -> bool:
    return dtype == torch.bfloat16 and bool(lib.lmv_mlp_fused_supported(C_, hidden, _lib.LMV_BF16))


def mlp_fused_fwd(xs: Sequence[Tensor], fc1: Folded, w2: Tensor, b2: Tensor, eps: float, scales: Optional[Sequence[Optional[Tensor]]] = None) -> List[Tensor]:
    """out_i = x_i + scale_i * fc2(GELU(fc1(LN(x_i)))) for up to two token matrices sharing the weights, ONE launch, hidden on chip."""
    C_ = xs[0].shape[-1]
    Hd = w2.shape[1]
    outs = [torch.empty_like(x) for x in xs]
    arr = (_lib.MlpProblem * len(xs))()
    for i, (s, x, o) in enumerate(zip(arr, xs, outs)):
        s.x, s.out, s.rows = _ptr(x), _ptr(o), x.numel() // C_
        sc = scales[i] if scales is not None else None
        s.row_scale, s.rows_per_sample = _f32(sc), (x.shape[1] if sc is not None else 0)
    w = _lib.MlpWeights(_ptr(fc1.wf), _f32(fc1.colsum), _f32(fc1.bf), _ptr(w2), _f32(b2))
    check(lib.lmv_mlp_fused_fwd(arr, len(xs), C.byref(w), C_, Hd, eps, dtype_code(xs[0]), _stream()), "lmv_mlp_fused_fwd")
    return outs


def attn_out_proj_residual(probs: Sequence[Prob], C_: int) -> None:
    """out = res + row_scale * (a @ W^T + b): attention output projection with the block's residual / DropPath in the epilogue."""
    check(lib.lmv_attn_out_proj_residual(_pack(probs), len(probs), C_, dtype_code(probs[0].a), _stream()), "lmv_attn_out_proj_residual")


class DwBatch:
    """Deferred split-K reductions (lmv_linear_dw_partial / lmv_reduce_batch): the weight-gradient GEMMs of a block leave their
    partial slabs in distinct regions of one arena; flush() sums them all in one launch."""

    def __init__(self):
        self.segs: List[_lib.ReduceSeg] = []
        self.keep: List[object] = []          # arenas (and operands) that must outlive the pending launches
        self.arena: Optional[Tensor] = None
        self.used = 0

    def alloc(self, nbytes: int, device, stream: int) -> int:
        nbytes = (nbytes + 255) // 256 * 256
        if self.arena is None or self.used + nbytes > self.arena.numel():
            if self.arena is not None:
                self.keep.append(self.arena)      # pending segments still point into it
            size = max(nbytes * 4, (self.arena.numel() * 2 if self.arena is not None else 1 << 26))
            self.arena = torch.empty(size, device=device, dtype=torch.uint8)
            self.arena.record_stream(torch.cuda.ExternalStream(stream, device=device))
            self.used = 0
        off = self.used
        self.used += nbytes
        return self.arena.data_ptr() + off

    def flush(self, stream: Optional[int] = None) -> None:
        if self.segs:
            arr = (_lib.ReduceSeg * len(self.segs))(*self.segs)
            check(lib.lmv_reduce_batch(arr, len(self.segs), _stream() if stream is None else stream), "lmv_reduce_batch")
        self.segs.clear(); self.keep.clear()
        self.used = 0


def linear_dw(probs: Sequence[Prob], N: int, K: int, stream: Optional[int] = None, batch: Optional[DwBatch] = None) -> None:
    """out (fp32 [N,K]) += a^T @ w ; bias_grad (fp32 [N]) += colsum(a); a = dY [rows,N], w = X [rows,K].
    stream: raw HIP stream handle to launch on (default: the current stream); the scratch is per stream.
    batch: defer the slab reduction to batch.flush() (same stream), which sums the slabs of every pending GEMM in one launch."""
    arr, code = _pack(probs), dtype_code(probs[0].a)
    st = _stream() if stream is None else stream
    if batch is not None:
        nb = lib.lmv_linear_dw_workspace_bytes(arr, len(probs), N, K, code)
        ws = batch.alloc(max(nb, 256), probs[0].a.device, st)
        segs = (_lib.ReduceSeg * 2)()
        n = C.c_int(0)
        check(lib.lmv_linear_dw_partial(arr, len(probs), N, K, ws, max(nb, 256), code, st, segs, C.byref(n)), "lmv_linear_dw_partial")
        for i in range(n.value):
            batch.segs.append(_lib.ReduceSeg.from_buffer_copy(segs[i]))
        return
    ws = _workspace(lib.lmv_linear_dw_workspace_bytes(arr, len(probs), N, K, code), probs[0].a.device, st)
    check(lib.lmv_linear_dw(arr, len(probs), N, K, ws.data_ptr(), ws.numel(), code, st), "lmv_linear_dw")


def reduce_segments(segs: Sequence["_lib.ReduceSeg"]) -> None:
    if segs:
        arr = (_lib.ReduceSeg * len(segs))(*segs)
        check(lib.lmv_reduce_batch(arr, len(segs), _stream()), "lmv_reduce_batch")


# -------------------------------------------------------------------------------------------
# BatchNorm2d (training mode) over channels-last rows
# -------------------------------------------------------------------------------------------
def batchnorm_train_fwd(x2d: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor],
                        momentum: float, eps: float, gelu: bool = False):
    """x2d [rows, C] (a channels-last feature map).  Returns (y, stats[2, C] = batch mean / rstd); running stats updated in place."""
    rows, C_ = x2d.shape
    y = torch.empty_like(x2d)
    stats = torch.empty((2, C_), device=x2d.device, dtype=torch.float32)
    ws = _workspace(lib.lmv_batchnorm_workspace_bytes(C_), x2d.device)
    check(lib.lmv_batchnorm_train_fwd(_ptr(x2d), _f32(gamma), _f32(beta), _f32(running_mean), _f32(running_var), momentum, eps, ACT_GELU if gelu else ACT_NONE,
                                      _ptr(y), _f32(stats), rows, C_, ws.data_ptr(), ws.numel(), dtype_code(x2d), _stream()), "lmv_batchnorm_train_fwd")
    return y, stats


def batchnorm_train_bwd(dy2d: Tensor, x2d: Tensor, gamma: Tensor, beta: Tensor, stats: Tensor, gelu: bool = False):
    """Returns (dx, dgamma, dbeta) of batchnorm_train_fwd."""
    rows, C_ = x2d.shape
    dx = torch.empty_like(x2d)
    dgamma = torch.empty((C_,), device=x2d.device, dtype=torch.float32)
    dbeta = torch.empty((C_,), device=x2d.device, dtype=torch.float32)
    ws = _workspace(lib.lmv_batchnorm_workspace_bytes(C_), x2d.device)
    check(lib.lmv_batchnorm_train_bwd(_ptr(dy2d), _ptr(x2d), _f32(gamma), _f32(beta), _f32(stats), ACT_GELU if gelu else ACT_NONE, _ptr(dx), _f32(dgamma),
                                      _f32(dbeta), rows, C_, ws.data_ptr(), ws.numel(), dtype_code(x2d), _stream()), "lmv_batchnorm_train_bwd")
    return dx, dgamma, dbeta


# -------------------------------------------------------------------------------------------
# LayerNorm
# -------------------------------------------------------------------------------------------
def layernorm_fwd_multi(xs: Sequence[Tensor], gamma: Tensor, beta: Tensor, eps: float, want_stats: bool = False, gelu: bool = False):
    """LayerNorm (gelu=True: GELU(LayerNorm)) of up to two tensors sharing (gamma, beta) in ONE launch; returns (ys, stats)."""
    C_ = xs[0].shape[-1]
    seg = (LnSegment * len(xs))()
    ys, sts = [], []
    for s, x in zip(seg, xs):
        rows = x.numel() // C_
        y = torch.empty_like(x)
        st = torch.empty((rows, 2), device=x.device, dtype=torch.float32) if want_stats else None
        s.x, s.y, s.stats, s.rows = _ptr(x), _ptr(y), _ptr(st), rows
        ys.append(y); sts.append(st)
    fn = lib.lmv_layernorm_gelu_fwd if gelu else lib.lmv_layernorm_fwd
    check(fn(seg, len(xs), _f32(gamma), _f32(beta), C_, eps, dtype_code(xs[0]), _stream()), "lmv_layernorm_fwd")
    return ys, sts


def layernorm_bwd_multi(dys: Sequence[Tensor], xs: Sequence[Tensor], stats: Sequence[Tensor], gamma: Tensor, dgamma: Tensor, dbeta: Tensor,
                        dres: Sequence[Optional[Tensor]], next_scales: Optional[Sequence[Optional[Tensor]]] = None, gelu_beta: Optional[Tensor] = None,
                        split_reduce: bool = False):
    """dx_i = dres_i + LN'(dy_i) for up to two tensors in ONE launch; dgamma / dbeta (fp32) are accumulated in place.
    next_scales: per-sample DropPath vectors of the NEXT backward stage; when given, returns (dxs, scaled) where scaled[i] is
    dx_i * next_scales[i][sample] written by the same launch (or dx_i itself where the scale is None)."""
    C_ = xs[0].shape[-1]
    seg = (LnSegment * len(xs))()
    dxs, scaled, total = [], [], 0
    for i, (s, dy, x, st, dr) in enumerate(zip(seg, dys, xs, stats, dres)):
        dx = torch.empty_like(x)
        s.x, s.dy, s.stats, s.dres, s.dx, s.rows = _ptr(x), _ptr(dy), _f32(st), _ptr(dr), _ptr(dx), x.numel() // C_
        sc = None if next_scales is None else next_scales[i]
        if sc is not None:
            d2 = torch.empty_like(x)
            s.dx_scale, s.dx_scaled, s.rows_per_sample = _f32(sc), _ptr(d2), x.shape[1]
            scaled.append(d2)
        else:
            scaled.append(dx)
        total += s.rows
        dxs.append(dx)
    code = dtype_code(xs[0])
    ws = _workspace(lib.lmv_layernorm_bwd_workspace_bytes(total, C_, code), xs[0].device)
    if split_reduce:               # the two-call form the native block scheduler uses (reduce on another stream there); same stream here
        rows = C.c_int(0)
        check(lib.lmv_layernorm_bwd_partial(seg, len(xs), _f32(gamma), C_, ws.data_ptr(), ws.numel(), C.byref(rows), code, _stream()), "lmv_layernorm_bwd_partial")
        check(lib.lmv_layernorm_bwd_reduce(ws.data_ptr(), rows.value, C_, _f32(dgamma), _f32(dbeta), _stream()), "lmv_layernorm_bwd_reduce")
        return dxs if next_scales is None else (dxs, scaled)
    if gelu_beta is not None:      # backward of GELU(LayerNorm(x)): needs beta to recompute the pre-activation
        check(lib.lmv_layernorm_gelu_bwd(seg, len(xs), _f32(gamma), _f32(gelu_beta), _f32(dgamma), _f32(dbeta), C_, ws.data_ptr(), ws.numel(), code,
                                         _stream()), "lmv_layernorm_gelu_bwd")
    else:
        check(lib.lmv_layernorm_bwd(seg, len(xs), _f32(gamma), _f32(dgamma), _f32(dbeta), C_, ws.data_ptr(), ws.numel(), code, _stream()),
              "lmv_layernorm_bwd")
    return dxs if next_scales is None else (dxs, scaled)


def ln_exact_fused(N: int, K: int, dtype) -> bool:
    """Does the block schedule run LayerNorm -> Linear as ONE launch (lmv_ln_linear_exact_fwd)?  Same rule as csrc/block.hip::ln_exact_ok."""
    return dtype == torch.bfloat16 and bool(lib.lmv_ln_linear_exact_fwd_supported(N, K, _lib.LMV_BF16)) and _lib.config_get("ln_exact_fused") != 0


def ln_linear_exact_fwd(probs: Sequence[Prob], N: int, K: int, gamma: Tensor, beta: Tensor, eps: float, want_stats: bool, want_ln: bool = True):
    """probs[i].a = raw rows: writes probs[i].out = LN(a) W^T + bias; returns (LayerNorm outputs, stats)  (lmv_ln_linear_exact_fwd).
    want_ln=False (inference): the normalised rows are not written (None in their place)."""
    arr = _pack(probs)
    seg = (LnSegment * len(probs))()
    ys, sts = [], []
    for s, p in zip(seg, probs):
        y = torch.empty_like(p.a) if want_ln else None
        st = torch.empty((p.rows, 2), device=p.a.device, dtype=torch.float32) if want_stats else None
        s.y, s.stats, s.rows = (_ptr(y) if want_ln else None), _f32(st), p.rows
        ys.append(y); sts.append(st)
    check(lib.lmv_ln_linear_exact_fwd(arr, seg, len(probs), N, K, _f32(gamma), _f32(beta), eps, dtype_code(probs[0].a), _stream()), "lmv_ln_linear_exact_fwd")
    return ys, sts


def res_ln_fused(C_: int, K: int, rows: int, dtype) -> bool:
    """Does the block schedule run `out = res + s (a W^T + b)` and the LayerNorm of `out` as ONE launch (lmv_linear_res_ln_fwd)?  The same
    rule as csrc/block.hip::res_ln_ok: supported shape, enough rows to fill the chip with 128-row panels, switch on."""
    return (dtype == torch.bfloat16 and 16384 <= rows <= 32768 and bool(lib.lmv_linear_res_ln_fwd_supported(C_, K, _lib.LMV_BF16)) and _lib.config_get("res_ln_fused") != 0)


def linear_res_ln_fwd(probs: Sequence[Prob], N: int, K: int, gamma: Tensor, beta: Tensor, eps: float, want_stats: bool):
    """probs as for linear_fwd (res required): writes p.out; returns (ys, stats) = LayerNorm of the outputs (lmv_linear_res_ln_fwd)."""
    arr = _pack(probs)
    seg = (LnSegment * len(probs))()
    ys, sts = [], []
    for s, p in zip(seg, probs):
        y = torch.empty_like(p.out)
        st = torch.empty((p.rows, 2), device=p.out.device, dtype=torch.float32) if want_stats else None
        s.y, s.stats, s.rows = _ptr(y), _f32(st), p.rows
        ys.append(y); sts.append(st)
    check(lib.lmv_linear_res_ln_fwd(arr, seg, len(probs), N, K, _f32(gamma), _f32(beta), eps, dtype_code(probs[0].a), _stream()), "lmv_linear_res_ln_fwd")
    return ys, sts


def linear_dx_ln_bwd(dys: Sequence[Tensor], wt: Tensor, xs: Sequence[Tensor], stats: Sequence[Tensor], gamma: Tensor, dgamma: Tensor, dbeta: Tensor,
                     dres: Sequence[Optional[Tensor]], next_scales: Optional[Sequence[Optional[Tensor]]] = None):
    """The dX of a Linear through its TRANSPOSED weight `wt` [C, N] fused with the LayerNorm backward of the Linear's input
    (lmv_linear_dx_ln_bwd; bf16, C = 384): dx_i = dres_i + LN'(dys_i @ wt^T); dgamma / dbeta accumulated; (dxs, scaled) as layernorm_bwd_multi."""
    C_, N = wt.shape
    seg = (LnSegment * len(xs))()
    pr = (LinearProblem * len(xs))()
    dxs, scaled, total = [], [], 0
    for i, (s, q, dy, x, st, dr) in enumerate(zip(seg, pr, dys, xs, stats, dres)):
        dx = torch.empty_like(x)
        s.x, s.stats, s.dres, s.dx, s.rows = _ptr(x), _f32(st), _ptr(dr), _ptr(dx), x.numel() // C_
        q.a, q.w, q.rows = _ptr(dy), _ptr(wt), s.rows
        sc = None if next_scales is None else next_scales[i]
        if sc is not None:
            d2 = torch.empty_like(x)
            s.dx_scale, s.dx_scaled, s.rows_per_sample = _f32(sc), _ptr(d2), x.shape[1]
            scaled.append(d2)
        else:
            scaled.append(dx)
        total += s.rows
        dxs.append(dx)
    code = dtype_code(xs[0])
    ws = _workspace(lib.lmv_linear_dx_ln_bwd_workspace_bytes(total, C_), xs[0].device)
    rows = C.c_int(0)
    check(lib.lmv_linear_dx_ln_bwd(pr, seg, len(xs), C_, N, _f32(gamma), ws.data_ptr(), ws.numel(), C.byref(rows), code, _stream()), "lmv_linear_dx_ln_bwd")
    check(lib.lmv_layernorm_bwd_reduce(ws.data_ptr(), rows.value, C_, _f32(dgamma), _f32(dbeta), _stream()), "lmv_layernorm_bwd_reduce")
    return dxs if next_scales is None else (dxs, scaled)


def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, want_stats: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    ys, sts = layernorm_fwd_multi([x], gamma, beta, eps, want_stats)
    return ys[0], sts[0]


def layernorm_bwd(dy: Tensor, x: Tensor, stats: Tensor, gamma: Tensor, dgamma: Tensor, dbeta: Tensor, dres: Optional[Tensor] = None) -> Tensor:
    """dx = dres + LN'(dy); dgamma / dbeta (fp32) are accumulated in place."""
    return layernorm_bwd_multi([dy], [x], [stats], gamma, dgamma, dbeta, [dres])[0]


# -------------------------------------------------------------------------------------------
# depth-wise 3x3 position embedding (token-major x: [B, H*W, C])
# -------------------------------------------------------------------------------------------
def dwconv_residual_fwd(x: Tensor, weight: Tensor, bias: Tensor, H: int, W: int) -> Tensor:
    B, N, C_ = x.shape
    assert N == H * W
    y = torch.empty_like(x)
    check(lib.lmv_dwconv3x3_residual_fwd(_ptr(x), _f32(weight), _f32(bias), _ptr(y), B, H, W, C_, dtype_code(x), _stream()),
          "lmv_dwconv3x3_residual_fwd")
    return y


def dwconv_residual_bwd_data(dy: Tensor, weight: Tensor, H: int, W: int) -> Tensor:
    B, N, C_ = dy.shape
    dx = torch.empty_like(dy)
    check(lib.lmv_dwconv3x3_residual_bwd_data(_ptr(dy), _f32(weight), _ptr(dx), B, H, W, C_, dtype_code(dy), _stream()),
          "lmv_dwconv3x3_residual_bwd_data")
    return dx


def dwconv_bwd_weight(dy: Tensor, x: Tensor, dweight: Tensor, dbias: Tensor, H: int, W: int, stream: Optional[int] = None) -> None:
    """stream: raw HIP stream handle to launch on (default: the current stream); the scratch is per stream."""
    B, N, C_ = dy.shape
    st = _stream() if stream is None else stream
    ws = _workspace(lib.lmv_dwconv3x3_bwd_weight_workspace_bytes(B, H, W, C_, dtype_code(dy)), dy.device, st)
    check(lib.lmv_dwconv3x3_bwd_weight(_ptr(dy), _ptr(x), _f32(dweight), _f32(dbias), B, H, W, C_, ws.data_ptr(), ws.numel(), dtype_code(dy), st),
          "lmv_dwconv3x3_bwd_weight")


# -------------------------------------------------------------------------------------------
# attention cores; q/k/v are (tensor, column offset) views into packed projections [B, L, X*C]
# -------------------------------------------------------------------------------------------
def _desc(q: Tuple[Tensor, int], k: Tuple[Tensor, int], v: Tuple[Tensor, int], o: Tensor, lse: Optional[Tensor], C_: int, scale: float) -> AttnDesc:
    qt, qo = q; kt, ko = k; vt, vo = v
    es = qt.element_size()
    d = AttnDesc()
    d.q, d.k, d.v = _ptr(qt) + qo * es, _ptr(kt) + ko * es, _ptr(vt) + vo * es
    d.o, d.lse = _ptr(o), _f32(lse)
    B, Lq, Lk = qt.shape[0], qt.shape[1], kt.shape[1]
    d.q_bs, d.q_rs = Lq * qt.shape[2], qt.shape[2]
    d.k_bs, d.k_rs = Lk * kt.shape[2], kt.shape[2]
    d.v_bs, d.v_rs = Lk * vt.shape[2], vt.shape[2]
    d.o_bs, d.o_rs = Lq * o.shape[2], o.shape[2]
    d.B, d.H, d.Lq, d.Lk, d.scale = B, C_ // HEAD_DIM, Lq, Lk, scale
    return d


def attn_fwd(q, k, v, C_: int, scale: float, want_lse: bool = False, stream: Optional[int] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """o[b,l,h,:] = softmax(scale q.k^T) v with q/k/v = (packed tensor [B,L,XC], column offset).
    stream: raw HIP stream handle to launch on (default: the current stream); the scratch is per stream."""
    qt = q[0]
    B, Lq, Lk = qt.shape[0], qt.shape[1], k[0].shape[1]
    o = torch.empty((B, Lq, C_), device=qt.device, dtype=qt.dtype)
    lse = torch.empty((B, C_ // HEAD_DIM, Lq), device=qt.device, dtype=torch.float32) if want_lse else None
    d = _desc(q, k, v, o, lse, C_, scale)
    nb = lib.lmv_attn_workspace_bytes(d.B, d.H, Lq, Lk, 0)
    st = _stream() if stream is None else stream
    ws = _workspace(nb, qt.device, st)
    check(lib.lmv_attn_fwd(C.byref(d), ws.data_ptr(), ws.numel(), dtype_code(qt), st), "lmv_attn_fwd")
    return o, lse


def attn_bwd(q, k, v, o: Tensor, lse: Tensor, d_o: Tensor, dq, dk, dv, C_: int, scale: float, stream: Optional[int] = None) -> None:
    """Writes dq/dk/dv = (packed grad tensor, column offset) with the layout of q/k/v."""
    d = _desc(q, k, v, o, lse, C_, scale)
    es = q[0].element_size()
    d.d_o = _ptr(d_o)
    d.dq, d.dk, d.dv = _ptr(dq[0]) + dq[1] * es, _ptr(dk[0]) + dk[1] * es, _ptr(dv[0]) + dv[1] * es
    nb = lib.lmv_attn_workspace_bytes(d.B, d.H, d.Lq, d.Lk, 1)
    st = _stream() if stream is None else stream
    ws = _workspace(nb, o.device, st)
    check(lib.lmv_attn_bwd(C.byref(d), ws.data_ptr(), ws.numel(), dtype_code(o), st), "lmv_attn_bwd")


def attn_fwd_pair(qkvs: Sequence[Tensor], C_: int, scale: float, want_lse: bool = False):
    """Self-attention of TWO packed qkv tensors [B, L_i, 3C] (same B and C) in one launch where the kernels allow (lmv_attn_fwd_pair);
    returns ([o_0, o_1], [lse_0, lse_1])."""
    descs = (AttnDesc * 2)()
    outs, lses, nb = [], [], 256
    for i, t in enumerate(qkvs):
        B, L = t.shape[0], t.shape[1]
        o = torch.empty((B, L, C_), device=t.device, dtype=t.dtype)
        lse = torch.empty((B, C_ // HEAD_DIM, L), device=t.device, dtype=torch.float32) if want_lse else None
        descs[i] = _desc((t, 0), (t, C_), (t, 2 * C_), o, lse, C_, scale)
        nb = max(nb, lib.lmv_attn_workspace_bytes(B, C_ // HEAD_DIM, L, L, 0))
        outs.append(o); lses.append(lse)
    ws = _workspace(nb, qkvs[0].device)
    check(lib.lmv_attn_fwd_pair(descs, ws.data_ptr(), ws.numel(), dtype_code(qkvs[0]), _stream()), "lmv_attn_fwd_pair")
    return outs, lses


def attn_bwd_pair(qkvs: Sequence[Tensor], os_: Sequence[Tensor], lses: Sequence[Tensor], d_os: Sequence[Tensor], dqkvs: Sequence[Tensor], C_: int,
                  scale: float) -> None:
    """Backward of attn_fwd_pair: dqkvs[i] (packed like qkvs[i]) is written."""
    descs = (AttnDesc * 2)()
    nb = 256
    for i, (t, o, l, g, dq) in enumerate(zip(qkvs, os_, lses, d_os, dqkvs)):
        d = _desc((t, 0), (t, C_), (t, 2 * C_), o, l, C_, scale)
        es = t.element_size()
        d.d_o = _ptr(g)
        d.dq, d.dk, d.dv = _ptr(dq), _ptr(dq) + C_ * es, _ptr(dq) + 2 * C_ * es
        descs[i] = d
        nb = max(nb, lib.lmv_attn_workspace_bytes(d.B, d.H, d.Lq, d.Lk, 1))
    ws = _workspace(nb, qkvs[0].device)
    check(lib.lmv_attn_bwd_pair(descs, ws.data_ptr(), ws.numel(), dtype_code(qkvs[0]), _stream()), "lmv_attn_bwd_pair")


def dca_scales(N: int, M: int, C_: int) -> Tuple[float, float]:
    """models/lemevit.py:235,255-256."""
    base = C_ ** (-0.5)
    return math.log(M, N) * base, math.log(N, N) * base


SDPA_SCALE = HEAD_DIM ** (-0.5)


# -------------------------------------------------------------------------------------------
# utilities
# -------------------------------------------------------------------------------------------
def cast(src: Tensor, dtype: torch.dtype) -> Tensor:
    dst = torch.empty(src.shape, device=src.device, dtype=dtype)
    check(lib.lmv_cast(_ptr(src), dtype_code(src), _ptr(dst), dtype_code(dst), src.numel(), _stream()), "lmv_cast")
    return dst


def im2col3x3s2_c3(x: Tensor, dtype: torch.dtype) -> Tensor:
    """[B, 3, H, W] images (any strides, fp32 or bf16) -> [B * ceil(H/2) * ceil(W/2), 32] patch matrix of the stem's first conv."""
    B, C, H, W = x.shape
    if C != 3:
        raise ValueError("im2col3x3s2_c3: 3 input channels expected")
    out = torch.empty(B * ((H + 1) // 2) * ((W + 1) // 2), 32, device=x.device, dtype=dtype)
    sb, sc, sh, sw = x.stride()
    if not x.is_cuda:
        raise RuntimeError("lemevit_amd: tensors must be on the GPU (no CPU fallback exists)")
    check(lib.lmv_im2col3x3s2_c3(x.data_ptr(), dtype_code(x), _ptr(out), dtype_code(out), B, H, W, sb, sc, sh, sw, _stream()), "lmv_im2col3x3s2_c3")
    return out


def im2col3x3s2_nhwc(x: Tensor, KP: int) -> Tensor:
    """x [B, H, W, C] (NHWC, contiguous) -> [B * ceil(H/2) * ceil(W/2), KP] patch matrix of a 3x3 / stride-2 / pad-1 convolution."""
    B, H, W, C_ = x.shape
    out = torch.empty(B * ((H + 1) // 2) * ((W + 1) // 2), KP, device=x.device, dtype=x.dtype)
    check(lib.lmv_im2col3x3s2_nhwc(_ptr(x), _ptr(out), B, H, W, C_, KP, dtype_code(x), _stream()), "lmv_im2col3x3s2_nhwc")
    return out


def conv3x3s2_implicit_ok(x: Tensor, Cout: int, KP: int) -> bool:
    """Whether lmv_conv3x3s2_fwd / _dw take this NHWC map (bf16, whole k-tiles): else the patch-matrix form (im2col3x3s2_nhwc + linear_*)."""
    B, H, W, C_ = x.shape
    rows = B * ((H + 1) // 2) * ((W + 1) // 2)
    return (x.is_cuda and x.dtype == torch.bfloat16 and C_ % 8 == 0 and Cout % 8 == 0 and KP % 64 == 0 and KP >= 9 * C_ and rows % 64 == 0 and rows < (1 << 22) and KP < (1 << 13)
            and x.numel() < (1 << 31))


def conv3x3s2_fwd(x: Tensor, wm: Tensor, bias: Optional[Tensor], act: int = ACT_NONE) -> Tensor:
    """Conv2d(Cin, Cout, 3, stride 2, padding 1) of the NHWC map x [B, H, W, Cin] as an implicit GEMM (no patch matrix): -> [B * Ho * Wo, Cout].
    wm [Cout, KP]: column (ky * 3 + kx) * Cin + ci (model._conv_matrix)."""
    B, H, W, C_ = x.shape
    Co, KP = wm.shape
    y = torch.empty(B * ((H + 1) // 2) * ((W + 1) // 2), Co, device=x.device, dtype=x.dtype)
    check(lib.lmv_conv3x3s2_fwd(_ptr(x), _ptr(wm), None if bias is None else _f32(bias), _ptr(y), B, H, W, C_, Co, KP, act, dtype_code(x), _stream()), "lmv_conv3x3s2_fwd")
    return y


def conv3x3s2_dw(dy: Tensor, x: Tensor, dwm: Tensor, dbias: Optional[Tensor]) -> None:
    """dwm [Cout, KP] += dy^T patches(x), dbias += column sums of dy (fp32, accumulated) -- the weight gradient of conv3x3s2_fwd, gathered from the map."""
    B, H, W, C_ = x.shape
    Co, KP = dwm.shape
    st = _stream()
    ws = _workspace(lib.lmv_conv3x3s2_dw_workspace_bytes(B, H, W, C_, Co, KP, dtype_code(x)), x.device, st)
    check(lib.lmv_conv3x3s2_dw(_ptr(dy), _ptr(x), _f32(dwm), None if dbias is None else _f32(dbias), B, H, W, C_, Co, KP, ws.data_ptr(), ws.numel(), dtype_code(x), st), "lmv_conv3x3s2_dw")


def col2im3x3s2_nhwc(dpatches: Tensor, B: int, H: int, W: int, C_: int) -> Tensor:
    """Gradient of im2col3x3s2_nhwc: [B * Ho * Wo, KP] -> [B, H, W, C]."""
    dx = torch.empty((B, H, W, C_), device=dpatches.device, dtype=dpatches.dtype)
    check(lib.lmv_col2im3x3s2_nhwc(_ptr(dpatches), _ptr(dx), B, H, W, C_, dpatches.shape[1], dtype_code(dpatches), _stream()), "lmv_col2im3x3s2_nhwc")
    return dx


def token_mean2_fwd(x: Tensor, c: Optional[Tensor]) -> Tensor:
    """out[b, :] = mean_l x[b, l, :] (+ mean_m c[b, m, :]) for token-major x [B, L, C], c [B, M, C]."""
    B, L, C_ = x.shape
    out = torch.empty((B, C_), device=x.device, dtype=x.dtype)
    check(lib.lmv_token_mean2_fwd(_ptr(x), L, _ptr(c), 0 if c is None else c.shape[1], C_, B, _ptr(out), dtype_code(x), _stream()), "lmv_token_mean2_fwd")
    return out


def token_mean2_affine_fwd(x: Tensor, c: Optional[Tensor], xscale: Tensor, xshift: Tensor) -> Tensor:
    """out[b, :] = xscale * mean_l x[b, l, :] + xshift (+ mean_m c[b, m, :]): the pool with an eval-mode BatchNorm folded in (fp32 scale / shift [C])."""
    B, L, C_ = x.shape
    if xscale.dtype != torch.float32 or xshift.dtype != torch.float32 or xscale.numel() != C_ or xshift.numel() != C_:
        raise TypeError("token_mean2_affine_fwd: xscale / xshift must be float32 [C]")
    out = torch.empty((B, C_), device=x.device, dtype=x.dtype)
    check(lib.lmv_token_mean2_affine_fwd(_ptr(x), L, _ptr(c), 0 if c is None else c.shape[1], C_, B, _ptr(xscale), _ptr(xshift), _ptr(out), dtype_code(x), _stream()),
          "lmv_token_mean2_affine_fwd")
    return out


def token_mean2_bwd(g: Tensor, L: int, M: int) -> Tuple[Tensor, Optional[Tensor]]:
    """Gradient of token_mean2_fwd: dx[b, l, :] = g[b, :] / L, dc[b, m, :] = g[b, :] / M (M = 0: no second segment)."""
    B, C_ = g.shape
    dx = torch.empty((B, L, C_), device=g.device, dtype=g.dtype)
    dc = torch.empty((B, M, C_), device=g.device, dtype=g.dtype) if M else None
    check(lib.lmv_token_mean2_bwd(_ptr(g), _ptr(dx), L, _ptr(dc), M, C_, B, dtype_code(g), _stream()), "lmv_token_mean2_bwd")
    return dx, dc


def row_scale(x: Tensor, scale: Tensor, rows_per_sample: int) -> Tensor:
    y = torch.empty_like(x)
    C_ = x.shape[-1]
    check(lib.lmv_row_scale(_ptr(x), _f32(scale), _ptr(y), x.numel() // C_, C_, rows_per_sample, dtype_code(x), _stream()), "lmv_row_scale")
    return y


def row_scale_multi(xs: Sequence[Tensor], scales: Sequence[Optional[Tensor]]) -> List[Tensor]:
    """DropPath scaling of up to two [B, L, C] tensors (same C) in ONE launch; a None scale passes its tensor through."""
    todo = [(x, s) for x, s in zip(xs, scales) if s is not None]
    if not todo:
        return list(xs)
    if len(todo) > 2 or len({x.shape[-1] for x, _ in todo}) != 1 or len({x.dtype for x, _ in todo}) != 1:
        return [x if s is None else row_scale(x, s, x.shape[1]) for x, s in zip(xs, scales)]
    from ._lib import RowScaleSegment
    arr = (RowScaleSegment * len(todo))()
    outs = []
    for seg, (x, s) in zip(arr, todo):
        y = torch.empty_like(x)
        seg.x, seg.scale, seg.y = _ptr(x), _f32(s), _ptr(y)
        seg.rows, seg.rows_per_sample = x.numel() // x.shape[-1], x.shape[1]
        outs.append(y)
    check(lib.lmv_row_scale_multi(arr, len(todo), todo[0][0].shape[-1], dtype_code(todo[0][0]), _stream()), "lmv_row_scale_multi")
    it = iter(outs)
    return [x if s is None else next(it) for x, s in zip(xs, scales)]


def transpose_batch(pairs: Sequence[Tuple[Tensor, Tensor]]) -> None:
    """dst = src^T for every (src [R, C], dst [C, R]) pair of contiguous bf16 matrices, in one launch per 48 pairs."""
    if not pairs:
        return
    arr = (_lib.TransposeSeg * len(pairs))()
    for sg, (src, dst) in zip(arr, pairs):
        if src.dtype != torch.bfloat16 or dst.dtype != torch.bfloat16 or src.dim() != 2 or dst.shape != (src.shape[1], src.shape[0]) \
                or not src.is_contiguous() or not dst.is_contiguous():
            raise TypeError("transpose_batch: contiguous bfloat16 [R, C] -> [C, R] pairs")
        sg.src, sg.dst, sg.rows, sg.cols = _ptr(src), _ptr(dst), src.shape[0], src.shape[1]
    check(lib.lmv_transpose_batch(arr, len(pairs), _lib.LMV_BF16, _stream()), "lmv_transpose_batch")


def adamw_flat(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, wd_mask: Optional[Tensor], lr: float, beta1: float,
               beta2: float, eps: float, weight_decay: float, step: int, shadow: Optional[Tensor] = None, step_dev: Optional[Tensor] = None) -> None:
    if shadow is not None and (shadow.dtype != torch.bfloat16 or shadow.numel() != param.numel()):
        raise TypeError("adamw_flat: shadow must be a bfloat16 tensor of the parameter buffer's length")
    if step_dev is not None and step_dev.dtype != torch.int32:
        raise TypeError("adamw_flat: step_dev must be an int32 device scalar")
    check(lib.lmv_adamw_flat(_f32(param), _f32(grad), _f32(exp_avg), _f32(exp_avg_sq), _f32(wd_mask), _ptr(shadow), param.numel(), lr, beta1, beta2, eps,
                             weight_decay, step, _ptr(step_dev), _stream()), "lmv_adamw_flat")


def ema_flat(ema: Tensor, param: Tensor, decay: float) -> None:
    """ema <- decay * ema + (1 - decay) * param over two flat fp32 buffers, one launch (lmv_ema_flat)."""
    if ema.dtype != torch.float32 or param.dtype != torch.float32 or ema.numel() != param.numel():
        raise TypeError("lemevit_amd: ema_flat takes two fp32 buffers of the same length")
    check(lib.lmv_ema_flat(_ptr(ema), _ptr(param), ema.numel(), float(decay), _stream()), "lmv_ema_flat")


# -------------------------------------------------------------------------------------------
# A run of "S" blocks as one persistent launch (csrc/sstage.hip; inference, bf16)
# -------------------------------------------------------------------------------------------
def sstage_supported(C_: int, heads: int, hidden: int, H: int, W: int, M: int, dtype: torch.dtype) -> bool:
    if dtype != torch.bfloat16:
        return False
    return bool(lib.lmv_sstage_supported(C_, heads, hidden, H, W, M, _lib.LMV_BF16))


class SStagePacked:
    """The parameters of `nblocks` consecutive S blocks in the layout lmv_sstage_fwd reads (lmv_sstage_pack)."""
    __slots__ = ("wpk", "vec", "nblocks", "C", "heads", "hidden", "layout")          # layout: "sstage" (lmv_sstage_pack) | "dstage" (lmv_dstage_pack): the two kernels' fragment orders differ


def _check_packed(P: "SStagePacked", want: str, C_: int, fn: str) -> None:
    """A pack built for the other stage kernel (or another width) would be read past its end / in the wrong fragment order: refuse it."""
    if getattr(P, "layout", None) != want or P.C != C_:
        raise ValueError(f"lemevit_amd: {fn} got parameters packed as {getattr(P, 'layout', None)!r} for C = {P.C}; it needs the {want!r} layout for C = {C_}")


SSTAGE_NAMES = ("attn.qkv.weight", "attn.proj.weight", "mlp.0.weight", "mlp.3.weight", "norm1.weight", "norm1.bias", "attn.qkv.bias", "attn.proj.bias",
                "norm2.weight", "norm2.bias", "mlp.0.bias", "mlp.3.bias", "pos_embed.weight", "pos_embed.bias")


def sstage_pack(blocks: Sequence[dict], heads: int) -> SStagePacked:
    """blocks: per block a dict name -> tensor (SSTAGE_NAMES; matrices bf16, vectors fp32, reference layouts, on the GPU)."""
    w0 = blocks[0]["attn.qkv.weight"]
    C_ = w0.shape[1]
    hidden = blocks[0]["mlp.0.weight"].shape[0]
    wb, vf = int(lib.lmv_sstage_wpk_bytes(C_, hidden)), int(lib.lmv_sstage_vec_floats(C_, hidden))
    P = SStagePacked()
    P.nblocks, P.C, P.heads, P.hidden, P.layout = len(blocks), C_, heads, hidden, "sstage"
    P.wpk = torch.empty(len(blocks) * wb, device=w0.device, dtype=torch.uint8)
    P.vec = torch.empty(len(blocks) * vf, device=w0.device, dtype=torch.float32)
    keep = []
    for j, blk in enumerate(blocks):
        bp = _lib.SStageBlockParams()
        bp.C, bp.heads, bp.hidden = C_, heads, hidden
        for field, name in zip(("qkv_w", "proj_w", "fc1_w", "fc2_w"), SSTAGE_NAMES[:4]):
            t = blk[name]
            if t.dtype != torch.bfloat16:
                raise TypeError("lemevit_amd: sstage_pack takes bf16 matrices")
            t = t.contiguous(); keep.append(t)
            setattr(bp, field, _ptr(t))
        for field, name in zip(("n1_w", "n1_b", "qkv_b", "proj_b", "n2_w", "n2_b", "fc1_b", "fc2_b", "pos_w", "pos_b"), SSTAGE_NAMES[4:]):
            t = blk[name].detach().float().contiguous(); keep.append(t)
            setattr(bp, field, _ptr(t))
        check(lib.lmv_sstage_pack(C.byref(bp), P.wpk.data_ptr() + j * wb, P.vec.data_ptr() + j * vf * 4, _stream()), "lmv_sstage_pack")
    return P


def sstage_max_concurrent(C_: int) -> int:
    """sstage_fwd launches of this width that may be in flight on different streams at once (lmv_sstage_max_concurrent; include/lemevit_hip.h, "Residency")."""
    return int(lib.lmv_sstage_max_concurrent(C_))


def _check_concurrent(concurrent: int, limit: int, fn: str) -> None:
    if concurrent > limit:
        raise RuntimeError(f"lemevit_amd: {fn} asked to run as one of {concurrent} concurrent launches, but this shape may have at most {limit} in flight on the device "
                           "(their incomplete slots would fill the chip and the in-launch hand-offs could starve): use the per-block schedule or fewer sub-batches")


def sstage_fwd(x: Tensor, c: Tensor, P: SStagePacked, H: int, W: int, eps: float, timing: Optional[Tensor] = None, timing_block: int = 0,
               concurrent: int = 1) -> Tuple[Tensor, Tensor]:
    """concurrent: how many stage launches the caller keeps in flight on the device at once, this one included (graph.split_forward's sub-batches)."""
    B, N, C_ = x.shape
    _check_packed(P, "sstage", C_, "sstage_fwd")
    _check_concurrent(concurrent, sstage_max_concurrent(C_), "sstage_fwd")
    d = _lib.SStageDesc()
    d.B, d.H, d.W, d.M, d.C, d.heads, d.hidden, d.nblocks, d.dtype, d.eps = B, H, W, c.shape[1], C_, P.heads, P.hidden, P.nblocks, dtype_code(x), eps
    d.wpk, d.vec = P.wpk.data_ptr(), P.vec.data_ptr()
    d.timing, d.timing_block = (None if timing is None else timing.data_ptr()), timing_block
    xo, co = torch.empty_like(x), torch.empty_like(c)
    nbytes = int(lib.lmv_sstage_workspace_bytes(min(B, int(lib.lmv_sstage_max_images(C_))), C_))
    ws = _workspace(nbytes, x.device)
    check(lib.lmv_sstage_fwd(C.byref(d), _ptr(x), _ptr(c), _ptr(xo), _ptr(co), ws.data_ptr(), ws.numel(), _stream()), "lmv_sstage_fwd")
    return xo, co


# -------------------------------------------------------------------------------------------
# A run of "D" blocks as one persistent launch (csrc/dstage.hip; inference, bf16)
# -------------------------------------------------------------------------------------------
def dstage_supported(C_: int, heads: int, hidden: int, H: int, W: int, M: int, dtype: torch.dtype) -> bool:
    if dtype != torch.bfloat16:
        return False
    return bool(lib.lmv_dstage_supported(C_, heads, hidden, H, W, M, _lib.LMV_BF16))


def dstage_max_concurrent(C_: int, H: int, kind: int = 0) -> int:
    """dstage_fwd launches of this shape that may be in flight on different streams at once (co-residency of their workgroups, csrc/dstage.hip)."""
    return int(lib.lmv_dstage_max_concurrent(C_, H, kind))


DSTAGE_NAMES = ("attn.qkv1.weight", "attn.qkv2.weight", "attn.proj_x.weight", "attn.proj_c.weight", "mlp.0.weight", "mlp.3.weight",
                "norm1.weight", "norm1.bias", "attn.qkv1.bias", "attn.qkv2.bias", "attn.proj_x.bias", "attn.proj_c.bias",
                "norm2.weight", "norm2.bias", "mlp.0.bias", "mlp.3.bias", "pos_embed.weight", "pos_embed.bias")
_DSTAGE_FIELDS = ("qkv1_w", "qkv2_w", "projx_w", "projc_w", "fc1_w", "fc2_w", "n1_w", "n1_b", "qkv1_b", "qkv2_b", "projx_b", "projc_b", "n2_w", "n2_b", "fc1_b", "fc2_b",
                  "pos_w", "pos_b")


def dstage_pack(blocks: Sequence[dict], heads: int) -> SStagePacked:
    """blocks: per block a dict name -> tensor (DSTAGE_NAMES; matrices bf16, vectors fp32, reference layouts, on the GPU)."""
    w0 = blocks[0]["attn.qkv1.weight"]
    C_ = w0.shape[1]
    hidden = blocks[0]["mlp.0.weight"].shape[0]
    wb, vf = int(lib.lmv_dstage_wpk_bytes(C_, hidden)), int(lib.lmv_dstage_vec_floats(C_, hidden))
    if wb == 0:
        raise ValueError(f"lemevit_amd: dstage_pack does not support C = {C_}")
    P = SStagePacked()
    P.nblocks, P.C, P.heads, P.hidden, P.layout = len(blocks), C_, heads, hidden, "dstage"
    P.wpk = torch.empty(len(blocks) * wb, device=w0.device, dtype=torch.uint8)
    P.vec = torch.empty(len(blocks) * vf, device=w0.device, dtype=torch.float32)
    keep = []
    for j, blk in enumerate(blocks):
        bp = _lib.DStageBlockParams()
        bp.C, bp.heads, bp.hidden = C_, heads, hidden
        for k, (field, name) in enumerate(zip(_DSTAGE_FIELDS, DSTAGE_NAMES)):
            t = blk[name]
            if k < 6:
                if t.dtype != torch.bfloat16:
                    raise TypeError("lemevit_amd: dstage_pack takes bf16 matrices")
                t = t.contiguous()
            else:
                t = t.detach().float().contiguous()
            keep.append(t)
            setattr(bp, field, _ptr(t))
        check(lib.lmv_dstage_pack(C.byref(bp), P.wpk.data_ptr() + j * wb, P.vec.data_ptr() + j * vf * 4, _stream()), "lmv_dstage_pack")
    return P


CSTAGE_NAMES = ("attn.q.weight", "attn.kv.weight", "attn.proj.weight", "mlp.0.weight", "mlp.3.weight", "norm1.weight", "norm1.bias", "attn.q.bias", "attn.kv.bias", "attn.proj.bias",
                "norm2.weight", "norm2.bias", "mlp.0.bias", "mlp.3.bias", "pos_embed.weight", "pos_embed.bias")


def cstage_pack(blocks: Sequence[dict], heads: int) -> SStagePacked:
    """A run of "C" blocks (CrossAttention, models/lemevit.py:421-498: q from the meta tokens, k / v from the image tokens) in the layout of the D-stage kernel:
    qkv1 = [0 | attn.kv] (the image tokens' k / v rows), qkv2 = [attn.q | 0 | 0] (the meta tokens' queries), proj_c = attn.proj; for dstage_fwd(kind=1)."""
    out = []
    for blk in blocks:
        kv, q = blk["attn.kv.weight"], blk["attn.q.weight"]
        C_ = q.shape[0]
        z = torch.zeros_like(q)
        zb = torch.zeros(C_, device=q.device, dtype=torch.float32)
        d = {n: blk[n] for n in ("mlp.0.weight", "mlp.3.weight", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "mlp.0.bias", "mlp.3.bias", "pos_embed.weight", "pos_embed.bias")}
        d["attn.qkv1.weight"] = torch.cat([z, kv], 0)
        d["attn.qkv2.weight"] = torch.cat([q, z, z], 0)
        d["attn.qkv1.bias"] = torch.cat([zb, blk["attn.kv.bias"].float()], 0)
        d["attn.qkv2.bias"] = torch.cat([blk["attn.q.bias"].float(), zb, zb], 0)
        d["attn.proj_x.weight"], d["attn.proj_x.bias"] = blk["attn.proj.weight"], blk["attn.proj.bias"]          # (not read)
        d["attn.proj_c.weight"], d["attn.proj_c.bias"] = blk["attn.proj.weight"], blk["attn.proj.bias"]
        out.append(d)
    return dstage_pack(out, heads)


def s2stage_pack(blocks: Sequence[dict], heads: int) -> SStagePacked:
    """A run of "S" blocks (SSTAGE_NAMES) in the layout of the D-stage kernel, for dstage_fwd(kind=2): the image tokens and the meta tokens share attn.qkv / attn.proj
    (qkv1 = qkv2, proj_x = proj_c).  Stage 3 of LeMeViT-Base at 384 x 384 (576 + 16 tokens: too long for sstage_fwd's two workgroups per image)."""
    out = []
    for blk in blocks:
        d = {n: blk[n] for n in ("mlp.0.weight", "mlp.3.weight", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "mlp.0.bias", "mlp.3.bias", "pos_embed.weight", "pos_embed.bias")}
        d["attn.qkv1.weight"] = d["attn.qkv2.weight"] = blk["attn.qkv.weight"]
        d["attn.qkv1.bias"] = d["attn.qkv2.bias"] = blk["attn.qkv.bias"]
        d["attn.proj_x.weight"] = d["attn.proj_c.weight"] = blk["attn.proj.weight"]
        d["attn.proj_x.bias"] = d["attn.proj_c.bias"] = blk["attn.proj.bias"]
        out.append(d)
    return dstage_pack(out, heads)


D2STAGE_NAMES = ("attn.qv1.weight", "attn.kv2.weight", "attn.proj_x.weight", "attn.proj_c.weight", "mlp.0.weight", "mlp.3.weight", "norm1.weight", "norm1.bias", "attn.qv1.bias", "attn.kv2.bias",
                 "attn.proj_x.bias", "attn.proj_c.bias", "norm2.weight", "norm2.bias", "mlp.0.bias", "mlp.3.bias", "pos_embed.weight", "pos_embed.bias")


def d2stage_pack(blocks: Sequence[dict], heads: int) -> SStagePacked:
    """A run of "D2" blocks (DualCrossAttention_v2, models/lemevit.py:327-418: the image tokens' q is also their key, the meta tokens' k also their query) in the D-stage
    kernel's layout, for dstage_fwd(kind=0): qkv1 = [q | q | v1] from attn.qv1, qkv2 = [k | k | v2] from attn.kv2 -- x' = softmax(q k^T s_x) v2, c' = softmax(k q^T s_c) v1."""
    out = []
    for blk in blocks:
        qv1, kv2 = blk["attn.qv1.weight"], blk["attn.kv2.weight"]
        C_ = qv1.shape[1]
        b1, b2 = blk["attn.qv1.bias"].float(), blk["attn.kv2.bias"].float()
        d = {n: blk[n] for n in ("attn.proj_x.weight", "attn.proj_c.weight", "attn.proj_x.bias", "attn.proj_c.bias", "mlp.0.weight", "mlp.3.weight", "norm1.weight", "norm1.bias",
                                 "norm2.weight", "norm2.bias", "mlp.0.bias", "mlp.3.bias", "pos_embed.weight", "pos_embed.bias")}
        d["attn.qkv1.weight"] = torch.cat([qv1[:C_], qv1[:C_], qv1[C_:]], 0)
        d["attn.qkv2.weight"] = torch.cat([kv2[:C_], kv2[:C_], kv2[C_:]], 0)
        d["attn.qkv1.bias"] = torch.cat([b1[:C_], b1[:C_], b1[C_:]], 0)
        d["attn.qkv2.bias"] = torch.cat([b2[:C_], b2[:C_], b2[C_:]], 0)
        out.append(d)
    return dstage_pack(out, heads)


def dstage_fwd(x: Tensor, c: Tensor, P: SStagePacked, H: int, W: int, eps: float, timing: Optional[Tensor] = None, timing_block: int = 0, kind: int = 0,
               concurrent: int = 1) -> Tuple[Tensor, Tensor]:
    """kind = 1: the packed blocks are "C" blocks (cstage_pack): x is returned as it came; kind = 2: "S" blocks (s2stage_pack).
    concurrent: how many stage launches the caller keeps in flight on the device at once, this one included."""
    B, N, C_ = x.shape
    _check_packed(P, "dstage", C_, "dstage_fwd")
    _check_concurrent(concurrent, dstage_max_concurrent(C_, H, kind), "dstage_fwd")
    d = _lib.SStageDesc()
    d.kind = kind
    d.B, d.H, d.W, d.M, d.C, d.heads, d.hidden, d.nblocks, d.dtype, d.eps = B, H, W, c.shape[1], C_, P.heads, P.hidden, P.nblocks, dtype_code(x), eps
    d.wpk, d.vec = P.wpk.data_ptr(), P.vec.data_ptr()
    d.timing, d.timing_block = (None if timing is None else timing.data_ptr()), timing_block
    xo, co = (x if kind == 1 else torch.empty_like(x)), torch.empty_like(c)
    ws = _workspace(int(lib.lmv_dstage_workspace_bytes(B, C_)), x.device)
    check(lib.lmv_dstage_fwd(C.byref(d), _ptr(x), _ptr(c), _ptr(xo), _ptr(co), ws.data_ptr(), ws.numel(), _stream()), "lmv_dstage_fwd")
    return xo, co


# -------------------------------------------------------------------------------------------
# The stem as one launch (csrc/stem.hip; inference, BatchNorm folded by the caller)
# -------------------------------------------------------------------------------------------
def stem_supported(H: int, W: int, Cm: int, Co: int, dtype: torch.dtype) -> bool:
    return dtype == torch.bfloat16 and bool(lib.lmv_stem_supported(H, W, Cm, Co, _lib.LMV_BF16))


def stem_pack(w1m: Tensor, w2m: Tensor) -> Tensor:
    """w1m [Cm, 32], w2m [Co, >= 9 Cm] bf16 (the GEMM operands of the two im2col forms) -> the packed weights of stem_fwd."""
    Cm, Co = w1m.shape[0], w2m.shape[0]
    if w1m.dtype != torch.bfloat16 or w2m.dtype != torch.bfloat16 or w1m.shape[1] != 32 or not w1m.is_contiguous() or w2m.stride(1) != 1:
        raise TypeError("lemevit_amd: stem_pack takes the bf16 [Cm, 32] and [Co, KP] conv matrices")
    nb = int(lib.lmv_stem_wpk_bytes(Cm, Co))
    if nb == 0:
        raise ValueError(f"lemevit_amd: stem_pack does not support {Cm} -> {Co} channels")
    out = torch.empty(nb, device=w1m.device, dtype=torch.uint8)
    check(lib.lmv_stem_pack(_ptr(w1m), _ptr(w2m), w2m.stride(0), Cm, Co, out.data_ptr(), _stream()), "lmv_stem_pack")
    return out


def stem_fwd(x: Tensor, wpk: Tensor, b1: Tensor, b2: Tensor, Cm: int, Co: int) -> Tensor:
    """x [B, 3, H, W] (any strides, fp32 / bf16) -> [B, H/4, W/4, Co] bf16."""
    B, C3, H, W = x.shape
    if C3 != 3 or not x.is_cuda:
        raise ValueError("lemevit_amd: stem_fwd takes [B, 3, H, W] images on the GPU")
    y = torch.empty((B, H // 4, W // 4, Co), device=x.device, dtype=torch.bfloat16)
    sb, sc, sh, sw = x.stride()
    check(lib.lmv_stem_fwd(x.data_ptr(), dtype_code(x), sb, sc, sh, sw, B, H, W, Cm, Co, wpk.data_ptr(), _f32(b1), _f32(b2), _ptr(y), _stream()), "lmv_stem_fwd")
    return y


def stage_error_count(reset: bool = False, sync: bool = True) -> int:
    """The sticky error word of the persistent stage kernels (a bounded in-launch wait that ran out = a lost hand-off).  It lives in pinned host memory: reading it costs nothing;
    sync=True synchronises the device first so that the answer covers every launch issued so far, sync=False covers the launches that have completed."""
    if sync:
        torch.cuda.synchronize()
    v = int(lib.lmv_stage_error_count(1 if reset else 0))
    if v < 0:
        check(v, "lmv_stage_error_count")
    return v


stage_kernels_disabled = False          # set by check_stage_errors: after a lost hand-off the process keeps to the per-block schedule (lemevit_amd/model.py::_sstage_applies)


def check_stage_errors(where: str, sync: bool = True) -> None:
    """Raise if a persistent stage kernel lost a hand-off (its outputs are then wrong).  Called at the synchronisation points of the callers -- bench.py after its timed
    regions, graph.try_graphed after its warm-up, LeMeViT.forward at entry (sync=False: the verdict on the PREVIOUS calls, free of charge) -- so that a lost hand-off is an
    exception, never a wrong tensor or a normal-looking bench line.  The word is cleared and the stage kernels are switched off for the rest of the process."""
    global stage_kernels_disabled
    n = stage_error_count(reset=True, sync=sync)
    if n:
        stage_kernels_disabled = True
        raise RuntimeError(f"lemevit_amd: a persistent stage kernel lost an in-launch hand-off ({where}): outputs of the stage launches since the last check are invalid. "
                           "Likely causes: another process on the device, more concurrent stage launches than lmv_*stage_max_concurrent allows, a device partition smaller than "
                           "the kernels were sized for.  The per-block schedule is used for the rest of this process.")
