"""Host-side mirror of the reference's model interface (models/lemevit.py) over the HIP kernels.

Same class names, constructor arguments, ``forward`` signatures and -- crucially -- the same
``state_dict()`` keys and shapes as the reference, so reference checkpoints load
unchanged and ``benchmark.py`` / ``main.py`` style callers (``create_model('lemevit_base')``,
``model(x)``, DDP, optimizers, autocast) work as before.  What differs is underneath:

* activations stay token-major [B, N, C] through a whole stage (no NCHW<->NLC bounce per block);
* every LeMeBlock is ONE autograd node whose forward and backward are schedules of hand-written
  gfx950 kernels (lemevit_amd/blocks.py);
* the batch-invariant meta-token prefix (``meta_tokens.repeat`` + ``meta_token_downsample[0]``,
  models/lemevit.py:812,833) is computed once and broadcast.

Boundary glue still served by PyTorch-ROCm library ops this round: the stem
and stride-2 down-sample conv+BatchNorm, the 16-token meta MLPs, and the BN/LN/mean-pool/head tail.

Compute dtype: fp32 input without autocast -> exact-fp32 kernels; bf16 input (``model.to(bfloat16)``)
or ``torch.autocast('cuda', torch.bfloat16)`` -> bf16 MFMA kernels with fp32 accumulation, fp32
statistics, fp32 master weights and gradients (the bf16 matrix copies are cached per parameter version).
"""
from __future__ import annotations

import os
import threading
import weakref
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .blocks import BLOCK_LN_EPS, PARAM_NAMES, block_backward, block_forward
from .ops import Prob

Tensor = torch.Tensor


def _cfg(url: str = "", **kwargs) -> dict:
    """timm.models.vision_transformer._cfg equivalent (models/lemevit.py:22,866)."""
    return dict(url=url, num_classes=1000, input_size=(3, 224, 224), pool_size=None, crop_pct=0.9, interpolation="bicubic",
                fixed_input_size=True, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                first_conv="patch_embed.proj", classifier="head", **kwargs)


# ------------------------------------------------------------------------------------------------
# compute-dtype copies of parameters
# ------------------------------------------------------------------------------------------------
_copy_cache: Dict[int, tuple] = {}


def _is_matrix(name: str) -> bool:
    return name.endswith(".weight") and (name.startswith("attn.") or name.startswith("mlp."))


_train_pass = 0          # bumped by every training-mode forward pass: casts made for an earlier pass are not reused


def new_training_pass() -> None:
    global _train_pass
    _train_pass += 1
    from . import blocks as _blocks
    _blocks.drain_deferred()        # a backward pass that raised leaves deferred side-stream joins (and its final callback) behind


def compute_copy(p: Tensor, want: torch.dtype) -> Tensor:
    """Detached tensor with p's values in dtype `want`.

    Inference: casts are cached per parameter version (load_state_dict / copy_ bump it).  Training: the cache is only
    valid within ONE forward pass -- fused optimizers (torch.optim.AdamW(fused=True), torch._fused_adamw_) update the
    parameters in place WITHOUT bumping ``_version``, so a version-keyed cache would keep feeding stale bf16 weights to
    the kernels.  An optimizer that maintains the bf16 copy itself (lemevit_amd.optim.FlatAdamW) attaches it as
    ``p._lmv_shadow`` and no cast is launched at all."""
    if p.dtype == want:
        return p.detach()
    shadow = getattr(p, "_lmv_shadow", None)
    if shadow is not None and shadow.dtype == want and shadow.device == p.device:
        return shadow
    key = id(p)
    ent = _copy_cache.get(key)
    stamp = (p._version, _train_pass, torch.is_grad_enabled() and p.requires_grad)      # a later training pass invalidates inference casts too
    if ent is not None and ent[0]() is p and ent[1] == stamp and ent[2].dtype == want and ent[2].device == p.device:
        return ent[2]
    global _casts_launched
    _casts_launched += 1             # a kernel on the current stream: image_ranges() re-forks its range streams behind it
    t = ops.cast(p.detach().contiguous(), want)
    _copy_cache[key] = (weakref.ref(p), stamp, t)
    return t


_casts_launched = 0


def _cache_filled() -> None:
    """A lazily built operand cache was (re)filled by kernels on the CURRENT stream: image_ranges() re-forks its range streams behind them and
    graph.split_forward orders its other sub-batch streams behind the sub-batch that filled it."""
    global _casts_launched
    _casts_launched += 1


def cache_fills() -> int:
    return _casts_launched


_fold_cache: dict = {}
_conv1_cache: dict = {}

# Inference: LayerNorm folded into the Linear that consumes it (ops.ln_fold -> lmv_ln_linear_fwd / lmv_mlp_fused_fwd).  The folded
# operands are cached per version of the four parameters involved (and per training pass, see compute_copy).
_FUSED = os.environ.get("LMV_FUSED", "1") != "0"          # 0: LayerNorm + Linear launches as in the training schedule (A/B runs)
_ln_fold_cache: dict = {}
_FOLD_PAIRS = {"S": (("attn.qkv", "norm1"),), "D": (("attn.qkv1", "norm1"), ("attn.qkv2", "norm1")), "C": (("attn.q", "norm1"), ("attn.kv", "norm1"))}


def _ln_fold_cached(w: Tensor, b: Optional[Tensor], g: Tensor, be: Tensor, dtype: torch.dtype) -> "ops.Folded":
    key = (id(w), dtype)
    stamp = tuple(t._version for t in (w, g, be)) + (None if b is None else b._version, _train_pass, id(g), id(be))
    ent = _ln_fold_cache.get(key)
    if ent is not None and ent[0]() is w and ent[1] == stamp:
        return ent[2]
    with torch.no_grad():
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        F = ops.ln_fold(f32(w), f32(b), f32(g), f32(be), dtype)
    _cache_filled()
    _ln_fold_cache[key] = (weakref.ref(w, lambda _r, k=key: _ln_fold_cache.pop(k, None)), stamp, F)      # dropped with the parameter (a deleted model does not leak its folds)
    return F


def block_folds(kind: str, params: "Dict[str, Tensor]", dtype: torch.dtype):
    """[(folded attention projections in slot order), folded mlp.0] of a S / D / C block, or None where the fused path does not apply."""
    if not _FUSED or dtype != torch.bfloat16:
        return None
    g1, b1, g2, b2 = params["norm1.weight"], params["norm1.bias"], params["norm2.weight"], params["norm2.bias"]
    attn = [_ln_fold_cached(params[n + ".weight"], params.get(n + ".bias"), g1, b1, dtype) for n, _ in _FOLD_PAIRS.get(kind, ())]      # "D2" / "Sx" (Python schedule): the MLP half only
    return attn, _ln_fold_cached(params["mlp.0.weight"], params.get("mlp.0.bias"), g2, b2, dtype)


def _conv1_matrix(weight: Tensor, dtype: torch.dtype) -> Tensor:
    """[Cout, 3, 3, 3] stem weight -> the [Cout, 32] GEMM operand of ops.im2col3x3s2_c3 (columns 27..31 zero), cached per version."""
    key = (id(weight), dtype)
    ent = _conv1_cache.get(key)
    stamp = (weight._version, _train_pass, torch.is_grad_enabled() and weight.requires_grad)      # see compute_copy
    if ent is not None and ent[0]() is weight and ent[1] == stamp:
        return ent[2]
    with torch.no_grad():
        m = torch.zeros(weight.shape[0], 32, device=weight.device, dtype=dtype)
        m[:, :27] = weight.detach().reshape(weight.shape[0], 27)
    _cache_filled()
    _conv1_cache[key] = (weakref.ref(weight, lambda _r, k=key: _conv1_cache.pop(k, None)), stamp, m)
    return m


class _StemConv1Fn(torch.autograd.Function):
    """First convolution of the stem (models/lemevit.py:713: 3 -> C/2 channels, 3x3, stride 2, padding 1) as im2col + the
    block GEMM kernels: y[NHWC] = patches @ W^T + b, dW = dY^T @ patches.  The image batch needs no gradient.  (MIOpen's
    implicit-GEMM kernel for this 27-deep reduction takes 455 us at B = 128; this is ~5x faster, weight gradient included.)"""

    @staticmethod
    def forward(ctx, x, weight, bias, cd, gelu=False):
        """gelu=True (inference only, BatchNorm folded into the weights): the GELU behind the first stem BatchNorm rides the GEMM epilogue."""
        B, _, H, W = x.shape
        Ho, Wo, Co = (H + 1) // 2, (W + 1) // 2, weight.shape[0]
        patches = ops.im2col3x3s2_c3(x, cd)
        y = torch.empty(B * Ho * Wo, Co, device=x.device, dtype=cd)
        b32 = None if bias is None else compute_copy(bias, torch.float32)
        ops.linear_fwd([Prob(patches, _conv1_matrix(weight, cd), y, bias=b32)], Co, 32, ops.ACT_GELU if gelu else ops.ACT_NONE)
        assert not (gelu and torch.is_grad_enabled() and weight.requires_grad), "the fused GELU epilogue has no backward"
        ctx.save_for_backward(patches)
        ctx.meta = (weight.shape, weight.dtype, None if bias is None else bias.dtype)
        return y.view(B, Ho, Wo, Co).permute(0, 3, 1, 2)            # NCHW-shaped, channels-last-strided: no copy

    @staticmethod
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        wshape, wdt, bdt = ctx.meta
        Co = wshape[0]
        g = dy.permute(0, 2, 3, 1).contiguous().view(-1, Co)
        if g.dtype != patches.dtype:
            g = g.to(patches.dtype)
        dwm = torch.zeros(Co, 32, device=g.device, dtype=torch.float32)
        db = torch.zeros(Co, device=g.device, dtype=torch.float32)
        ops.linear_dw([Prob(g, patches, dwm, bias_grad=db)], Co, 32)
        dw = dwm[:, :27].reshape(wshape).to(wdt)
        return None, dw, (None if bdt is None else db.to(bdt)), None, None


_conv_cache: dict = {}


def _conv_matrix(weight: Tensor, dtype: torch.dtype, KP: int) -> Tensor:
    """[Cout, Cin, 3, 3] weight -> the [Cout, KP] GEMM operand of ops.im2col3x3s2_nhwc: column (ky * 3 + kx) * Cin + ci, zero padding
    behind 9 Cin.  Cached like _conv1_matrix (per parameter version AND training pass)."""
    key = (id(weight), dtype, KP)
    ent = _conv_cache.get(key)
    stamp = (weight._version, _train_pass, torch.is_grad_enabled() and weight.requires_grad, weight.data_ptr())
    if ent is not None and ent[0]() is weight and ent[1] == stamp:
        return ent[2]
    with torch.no_grad():
        Co, Ci = weight.shape[0], weight.shape[1]
        m = torch.zeros(Co, KP, device=weight.device, dtype=dtype)
        m[:, :9 * Ci] = weight.detach().permute(0, 2, 3, 1).reshape(Co, 9 * Ci)
    # dropped with `weight`: the eval-mode conv + BatchNorm fold builds a NEW folded weight after every training pass, whose matrix would
    # otherwise stay cached under the dead tensor's id (~5 MB per train -> eval cycle for Base)
    _cache_filled()
    _conv_cache[key] = (weakref.ref(weight, lambda _r, k=key: _conv_cache.pop(k, None)), stamp, m)
    return m


class _Conv3x3s2Fn(torch.autograd.Function):
    """Conv2d(Cin, Cout, kernel 3, stride 2, padding 1) on a channels-last map as im2col + the block GEMM kernels (SURVEY section 8,
    row f1; models/lemevit.py:701-703, :714-717): y = patches @ Wm^T + b, dW = dY^T @ patches, dX = col2im(dY @ Wm).  No MIOpen
    solver search, no run-to-run differences in the gradients (the library's weight-gradient kernels are not reproducible)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cd):
        B, Ci, H, W = x.shape
        Co = weight.shape[0]
        xh = x.detach().permute(0, 2, 3, 1).to(cd).contiguous()            # NHWC; no copy for channels-last input of the right dtype
        KP = (9 * Ci + 63) // 64 * 64 if cd == torch.bfloat16 else 9 * Ci  # bf16: whole 64-deep k-steps (the GEMM's LDS-DMA path)
        Wm = _conv_matrix(weight, cd, KP)
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        b32 = None if bias is None else compute_copy(bias, torch.float32)
        implicit = _CONV_IMPLICIT and ops.conv3x3s2_implicit_ok(xh, Co, KP)
        if implicit:          # round 6: the GEMM gathers the patch elements from the map -- no patch matrix is written, read back, or kept for the backward pass
            y = ops.conv3x3s2_fwd(xh, Wm, b32)
            ctx.save_for_backward(xh, Wm)
        else:
            patches = ops.im2col3x3s2_nhwc(xh, KP)
            y = torch.empty(B * Ho * Wo, Co, device=x.device, dtype=cd)
            ops.linear_fwd([Prob(patches, Wm, y, bias=b32)], Co, KP)
            ctx.save_for_backward(patches, Wm)
        ctx.meta = (x.shape, x.dtype, weight.shape, weight.dtype, None if bias is None else bias.dtype, implicit)
        return y.view(B, Ho, Wo, Co).permute(0, 3, 1, 2)                    # NCHW-shaped, channels-last-strided: no copy

    @staticmethod
    def backward(ctx, dy):
        patches, Wm = ctx.saved_tensors
        (B, Ci, H, W), xdt, wshape, wdt, bdt, implicit = ctx.meta
        Co, KP = Wm.shape
        g = dy.permute(0, 2, 3, 1).contiguous().view(-1, Co)
        if g.dtype != patches.dtype:
            g = g.to(patches.dtype)
        dwm = torch.zeros(Co, KP, device=g.device, dtype=torch.float32)
        db = torch.zeros(Co, device=g.device, dtype=torch.float32)
        if implicit:
            ops.conv3x3s2_dw(g, patches, dwm, db)          # (`patches` is the NHWC map here)
        else:
            ops.linear_dw([Prob(g, patches, dwm, bias_grad=db)], Co, KP)
        dw = dwm[:, :9 * Ci].reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2).to(wdt)
        dx = None
        if ctx.needs_input_grad[0]:
            dp = torch.empty(g.shape[0], KP, device=g.device, dtype=g.dtype)
            ops.linear_dx([Prob(g, Wm, dp)], Co, KP)
            dx = ops.col2im3x3s2_nhwc(dp, B, H, W, Ci).permute(0, 3, 1, 2).to(xdt)
        return dx, dw, (None if bdt is None else db.to(bdt)), None


def _is_conv3x3s2(m: nn.Module, x: Tensor) -> bool:
    return (isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (2, 2) and m.padding == (1, 1) and m.dilation == (1, 1)
            and m.groups == 1 and m.in_channels % 8 == 0 and m.out_channels % 8 == 0 and m.padding_mode == "zeros" and x.is_cuda
            and x.dtype in (torch.float32, torch.bfloat16) and m.weight.dtype in (torch.float32, torch.bfloat16))          # (bf16 weights: benchmark.py's --precision bfloat16 whole-model cast)


_CONV_IMPLICIT = os.environ.get("LMV_CONV_IMPLICIT", "1") != "0"      # 0: the patch-matrix form of the 3 x 3 / stride-2 convolutions (im2col + GEMM; A/B runs)


class _BNActFn(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ the GELU behind the first stem BatchNorm) on a channels-last feature map
    (models/lemevit.py:698-704, 714-717, 773): lmv_batchnorm_train_fwd / _bwd instead of MIOpen's batch-norm kernels plus
    a separate GELU pass.  Running statistics are updated in place like nn.BatchNorm2d."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, gelu):
        B, C, H, W = x.shape
        x2d = x.detach().permute(0, 2, 3, 1).contiguous().view(B * H * W, C)        # no copy for channels-last input
        g32, b32 = compute_copy(weight, torch.float32), compute_copy(bias, torch.float32)
        y, stats = ops.batchnorm_train_fwd(x2d, g32, b32, running_mean, running_var, momentum, eps, gelu)
        ctx.save_for_backward(x2d, stats, g32, b32)
        ctx.meta = (gelu, (B, C, H, W), weight.dtype, bias.dtype)
        return y.view(B, H, W, C).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        x2d, stats, g32, b32 = ctx.saved_tensors
        gelu, (B, C, H, W), wdt, bdt = ctx.meta
        g = dy.permute(0, 2, 3, 1).contiguous().view(B * H * W, C)
        if g.dtype != x2d.dtype:
            g = g.to(x2d.dtype)
        dx, dgamma, dbeta = ops.batchnorm_train_bwd(g, x2d, g32, b32, stats, gelu)
        return dx.view(B, H, W, C).permute(0, 3, 1, 2), dgamma.to(wdt), dbeta.to(bdt), None, None, None, None, None



def _bn_native(m: nn.Module, x: Tensor) -> bool:
    """Can this BatchNorm2d call run on the HIP training kernels?"""
    return (isinstance(m, nn.BatchNorm2d) and m.training and m.affine and m.track_running_stats and m.momentum is not None and x.is_cuda
            and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % 8 == 0 and x.shape[1] <= 1024
            and x.shape[0] * x.shape[2] * x.shape[3] > 1 and m.weight.dtype == torch.float32)


def _bn_train(m: nn.BatchNorm2d, x: Tensor, gelu: bool = False) -> Tensor:
    if m.num_batches_tracked is not None:
        m.num_batches_tracked.add_(1)
    return _BNActFn.apply(x, m.weight, m.bias, m.running_mean, m.running_var, float(m.momentum), float(m.eps), gelu)


class _MetaMLPFn(torch.autograd.Function):
    """The per-stage meta-token MLP, Linear -> LayerNorm -> GELU -> Linear -> LayerNorm on [*, 16, C] (models/lemevit.py:731-743),
    as four launches of the block kernels (GEMM with bias epilogue, LayerNorm with fused GELU) and a hand-written backward --
    instead of ~25 library kernels and autocast casts per stage."""

    @staticmethod
    def forward(ctx, c, w1, b1, g1, be1, w2, b2, g2, be2, eps1, eps2, cd):
        cin, hid, cout = w1.shape[1], w1.shape[0], w2.shape[0]
        x = c.detach().to(cd).contiguous().view(-1, cin)
        R = x.shape[0]
        W1, W2 = compute_copy(w1, cd), compute_copy(w2, cd)
        f32 = lambda p: compute_copy(p, torch.float32)
        h1 = torch.empty(R, hid, device=x.device, dtype=cd)
        ops.linear_fwd([Prob(x, W1, h1, bias=f32(b1))], hid, cin)
        (a1,), (st1,) = ops.layernorm_fwd_multi([h1], f32(g1), f32(be1), eps1, want_stats=True, gelu=True)
        h2 = torch.empty(R, cout, device=x.device, dtype=cd)
        ops.linear_fwd([Prob(a1, W2, h2, bias=f32(b2))], cout, hid)
        (y,), (st2,) = ops.layernorm_fwd_multi([h2], f32(g2), f32(be2), eps2, want_stats=True)
        ctx.saved = (x, h1, st1, a1, h2, st2, W1, W2, f32(g1), f32(be1), f32(g2))
        ctx.meta = (c.shape, c.dtype, [(p.shape, p.dtype) for p in (w1, b1, g1, be1, w2, b2, g2, be2)])
        return y.view(c.shape[:-1] + (cout,))

    @staticmethod
    def backward(ctx, dy):
        x, h1, st1, a1, h2, st2, W1, W2, g1, be1, g2 = ctx.saved
        cshape, cdtype, pmeta = ctx.meta
        cin, hid, cout = W1.shape[1], W1.shape[0], W2.shape[0]
        dy = dy.contiguous().view(-1, cout)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        sizes = [int(torch.Size(sh).numel()) for sh, _ in pmeta]
        pad = [(n + 3) // 4 * 4 for n in sizes]
        flat = torch.zeros(sum(pad), device=x.device, dtype=torch.float32)
        G, off = [], 0
        for (sh, _), n, pd in zip(pmeta, sizes, pad):
            G.append(flat[off:off + n].view(sh)); off += pd
        dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2 = G
        (dh2,) = ops.layernorm_bwd_multi([dy], [h2], [st2], g2, dg2, dbe2, [None])
        ops.linear_dw([Prob(dh2, a1, dW2, bias_grad=db2)], cout, hid)
        da1 = torch.empty_like(a1)
        ops.linear_dx([Prob(dh2, W2, da1)], cout, hid)
        (dh1,) = ops.layernorm_bwd_multi([da1], [h1], [st1], g1, dg1, dbe1, [None], gelu_beta=be1)
        ops.linear_dw([Prob(dh1, x, dW1, bias_grad=db1)], hid, cin)
        dc = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ops.linear_dx([Prob(dh1, W1, dx)], hid, cin)
            dc = dx.view(cshape).to(cdtype)
        ctx.saved = None
        return (dc, *[g if dt == torch.float32 else g.to(dt) for g, (_, dt) in zip(G, pmeta)], None, None, None)


class _TailFn(torch.autograd.Function):
    """The classifier tail on the block kernels (models/lemevit.py:815-835):
    logits = head(mean_tokens(x) + mean_tokens(LayerNorm(c))) for token-major x [B, L, C] (already through the final BatchNorm)
    and the meta tokens c [B, M, C] -- lmv_layernorm_fwd -> lmv_token_mean2_fwd -> lmv_linear_fwd, and a hand-written backward
    (lmv_linear_dx / _dw, lmv_token_mean2_bwd, lmv_layernorm_bwd) instead of ~25 library kernels.  head = None returns the
    pooled features (forward_features)."""

    @staticmethod
    def forward(ctx, x, c, g, b, eps, hw, hb, cd):
        xc = x.detach().to(cd).contiguous()
        cc = c.detach().to(cd).contiguous()
        f32 = lambda p: compute_copy(p, torch.float32)
        (cn,), (st,) = ops.layernorm_fwd_multi([cc], f32(g), f32(b), eps, want_stats=True)
        pooled = ops.token_mean2_fwd(xc, cn)
        ctx.dims = (xc.shape[1], cc.shape[1])
        ctx.meta = (x.dtype, c.dtype, g.dtype, b.dtype, None if hw is None else hw.dtype, None if hb is None else hb.dtype, None if hw is None else hw.shape)
        if hw is None:
            ctx.saved = (cc, st, f32(g), None, None)
            return pooled
        W = compute_copy(hw, cd)
        N, K = W.shape
        Np = (N + 7) // 8 * 8                              # the GEMM wants a multiple of 8 output columns: pad the operand, slice the result
        if Np != N:
            Wp = torch.zeros((Np, K), device=W.device, dtype=cd); Wp[:N] = W
            bp = None
            if hb is not None:
                bp = torch.zeros((Np,), device=W.device, dtype=torch.float32); bp[:N] = f32(hb)
        else:
            Wp, bp = W, (None if hb is None else f32(hb))
        out = torch.empty((pooled.shape[0], Np), device=pooled.device, dtype=cd)
        ops.linear_fwd([Prob(pooled, Wp, out, bias=bp)], Np, K)
        ctx.saved = (cc, st, f32(g), pooled, Wp)
        return out[:, :N] if Np != N else out

    @staticmethod
    def backward(ctx, dout):
        cc, st, g32, pooled, Wp = ctx.saved
        L, M = ctx.dims
        xdt, cdt, gdt, bdt, hwdt, hbdt, hwshape = ctx.meta
        dW = db = None
        if Wp is not None:
            Np, K = Wp.shape
            N = hwshape[0]
            d = dout.contiguous()
            if d.dtype != pooled.dtype:
                d = d.to(pooled.dtype)
            if Np != N:
                dp = torch.zeros((d.shape[0], Np), device=d.device, dtype=d.dtype); dp[:, :N] = d
                d = dp
            dWf = torch.zeros((Np, K), device=d.device, dtype=torch.float32); dbf = torch.zeros((Np,), device=d.device, dtype=torch.float32)
            ops.linear_dw([Prob(d, pooled, dWf, bias_grad=dbf)], Np, K)
            dpooled = torch.empty_like(pooled)
            ops.linear_dx([Prob(d, Wp, dpooled)], Np, K)
            dW = dWf[:N].to(hwdt)
            db = None if hbdt is None else dbf[:N].to(hbdt)
        else:
            dpooled = dout.contiguous().to(cc.dtype)
        dx, dcn = ops.token_mean2_bwd(dpooled, L, M)
        dg = torch.zeros_like(g32); dbeta = torch.zeros_like(g32)
        (dc,) = ops.layernorm_bwd_multi([dcn], [cc], [st], g32, dg, dbeta, [None])
        ctx.saved = None
        return dx.to(xdt), dc.to(cdt), dg.to(gdt), dbeta.to(bdt), None, dW, db, None


_tail_cache: dict = {}


def _tail_infer(norm_c: nn.LayerNorm, bn: nn.BatchNorm2d, head: nn.Linear, xt: Tensor, c: Tensor, cd: torch.dtype) -> Tensor:
    """No-grad classifier tail (models/lemevit.py:815-835): logits = head(mean(BN_eval(x)) + mean(LayerNorm(c))).  The BatchNorm affine
    (scale, shift), the fp32 LayerNorm affine and the padded classifier operands are cached until one of their source tensors changes."""
    src = [bn.weight, bn.bias, bn.running_mean, bn.running_var, norm_c.weight, norm_c.bias, head.weight, head.bias]
    ver = (_train_pass, cd) + tuple(-1 if t is None else t._version for t in src) + tuple(0 if t is None else t.data_ptr() for t in src)
    key = (id(bn), id(head))
    ent = _tail_cache.get(key)
    if ent is None or ent[0] != ver:
        with torch.no_grad():
            a = torch.rsqrt(bn.running_var.float() + bn.eps)
            if bn.weight is not None:
                a = a * bn.weight.float()
            b = -bn.running_mean.float() * a
            if bn.bias is not None:
                b = b + bn.bias.float()
            N, K = head.weight.shape
            Np = (N + 7) // 8 * 8
            Wp = torch.zeros((Np, K), device=xt.device, dtype=cd); Wp[:N] = head.weight.detach().to(cd)
            bp = torch.zeros((Np,), device=xt.device, dtype=torch.float32)
            if head.bias is not None:
                bp[:N] = head.bias.detach().float()
            ent = (ver, a.contiguous(), b.contiguous(), norm_c.weight.detach().float().contiguous(), norm_c.bias.detach().float().contiguous(), Wp, bp, N)
        _cache_filled()
        _tail_cache[key] = ent
    _, a, b, g32, b32, Wp, bp, N = ent
    cc = c.detach().to(cd).contiguous()
    (cn,), _ = ops.layernorm_fwd_multi([cc], g32, b32, float(norm_c.eps), want_stats=False)
    pooled = ops.token_mean2_affine_fwd(xt.detach().to(cd).contiguous(), cn, a, b)
    Np, K = Wp.shape
    out = torch.empty((pooled.shape[0], Np), device=pooled.device, dtype=cd)
    ops.linear_fwd([Prob(pooled, Wp, out, bias=bp)], Np, K)
    return out[:, :N] if Np != N else out


def _tail_native(norm_c: nn.Module, head: Optional[nn.Module], xt: Tensor, c: Tensor, cd: torch.dtype) -> bool:
    ok = (isinstance(norm_c, nn.LayerNorm) and norm_c.elementwise_affine and norm_c.bias is not None and xt.is_cuda and cd in (torch.float32, torch.bfloat16)
          and xt.shape[-1] % 8 == 0 and norm_c.weight.dtype == torch.float32)
    if head is not None:
        ok = ok and isinstance(head, nn.Linear) and head.in_features % 8 == 0 and head.weight.dtype == torch.float32
    return ok


def _is_meta_mlp(seq: nn.Module, c: Tensor, cd: torch.dtype) -> bool:
    if not (isinstance(seq, nn.Sequential) and len(seq) == 5 and c.is_cuda and cd in (torch.float32, torch.bfloat16)):
        return False
    l1, n1, act, l2, n2 = seq
    return (isinstance(l1, nn.Linear) and isinstance(n1, nn.LayerNorm) and isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none"
            and isinstance(l2, nn.Linear) and isinstance(n2, nn.LayerNorm) and l1.bias is not None and l2.bias is not None
            and n1.elementwise_affine and n2.elementwise_affine and n1.bias is not None and n2.bias is not None
            and l1.in_features % 8 == 0 and l1.out_features % 8 == 0 and l2.out_features % 8 == 0 and l1.out_features <= 2048
            and l1.weight.dtype == torch.float32 and c.dtype in (torch.float32, torch.bfloat16))


def _is_stem_conv1(m: nn.Module, x: Tensor) -> bool:
    return (isinstance(m, nn.Conv2d) and m.in_channels == 3 and m.kernel_size == (3, 3) and m.stride == (2, 2) and m.padding == (1, 1)
            and m.dilation == (1, 1) and m.groups == 1 and m.out_channels % 8 == 0 and m.padding_mode == "zeros" and not x.requires_grad
            and x.dtype in (torch.float32, torch.bfloat16))


def _folded_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d, dtype: torch.dtype):
    """(weight, bias, fp32 bias) of conv followed by eval-mode BatchNorm, cached until any of the six source tensors changes."""
    src = [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    # _train_pass: fused optimizers and the native BatchNorm kernel update these tensors in place WITHOUT bumping
    # ``_version`` (see compute_copy), so a fold made before a training pass must not survive it
    ver = (_train_pass,) + tuple(-1 if t is None else t._version for t in src) + tuple(0 if t is None else t.data_ptr() for t in src)
    key = (id(conv), id(bn), dtype)
    ent = _fold_cache.get(key)
    if ent is not None and ent[0] == ver:
        return ent[1], ent[2], ent[3]
    with torch.no_grad():
        s = (bn.running_var.float() + bn.eps).rsqrt()
        if bn.weight is not None:
            s = s * bn.weight.float()
        w = conv.weight.float() * s[:, None, None, None]
        b0 = conv.bias.float() if conv.bias is not None else torch.zeros_like(s)
        b = (b0 - bn.running_mean.float()) * s
        if bn.bias is not None:
            b = b + bn.bias.float()
        w = w.to(dtype).contiguous(memory_format=torch.channels_last)
        b32 = b.contiguous()
        b = b.to(dtype)
    _cache_filled()
    _fold_cache[key] = (ver, w, b, b32)
    return w, b, b32


_STEM = os.environ.get("LMV_STEM", "1") != "0"          # 0: the stem as im2col + GEMM launches (A/B runs)
_stem_cache: dict = {}


def _stem_applies(mods, x: Tensor, cd: torch.dtype) -> bool:
    """models/lemevit.py:698-704: Conv(3, C/2, 3, 2, 1) - BN - GELU - Conv(C/2, C, 3, 2, 1) - BN in inference -> ONE launch (csrc/stem.hip)."""
    if not (_STEM and len(mods) == 5 and x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and cd == torch.bfloat16 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    c1, b1, act, c2, b2 = mods
    ok = lambda c, ci, co: (isinstance(c, nn.Conv2d) and c.kernel_size == (3, 3) and c.stride == (2, 2) and c.padding == (1, 1) and c.dilation == (1, 1) and c.groups == 1
                            and c.padding_mode == "zeros" and c.in_channels == ci and c.out_channels == co and c.weight.dtype == torch.float32)
    if not (isinstance(c1, nn.Conv2d) and isinstance(c2, nn.Conv2d) and ok(c1, 3, c1.out_channels) and ok(c2, c1.out_channels, c2.out_channels)):
        return False
    if not all(isinstance(b, nn.BatchNorm2d) and b.track_running_stats for b in (b1, b2)) or not (isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none"):
        return False
    return ops.stem_supported(x.shape[2], x.shape[3], c1.out_channels, c2.out_channels, cd)


def _stem_fused(mods, x: Tensor, cd: torch.dtype) -> Tensor:
    c1, b1, _, c2, b2 = mods
    w1, _, b1f = _folded_conv_bn(c1, b1, cd)
    w2, _, b2f = _folded_conv_bn(c2, b2, cd)
    key = (id(c1), id(c2))
    ent = _stem_cache.get(key)
    if ent is not None and ent[3]() is not c1:          # a recycled id: the module the entry was built for is gone
        ent = None
    if ent is None or ent[0] is not w1 or ent[1] is not w2:          # (the folds are new tensors whenever a source tensor changed)
        Cm = w1.shape[0]
        w1m = torch.zeros(Cm, 32, device=w1.device, dtype=cd)
        w1m[:, :27] = w1.reshape(Cm, 27)
        w2m = w2.permute(0, 2, 3, 1).reshape(w2.shape[0], 9 * Cm).contiguous()
        ent = (w1, w2, ops.stem_pack(w1m, w2m), weakref.ref(c1, lambda _r, k=key: _stem_cache.pop(k, None)))          # evicted with the module, as _conv_cache / _sstage_cache
        _cache_filled()
        _stem_cache[key] = ent
    y = ops.stem_fwd(x, ent[2], b1f, b2f, w1.shape[0], w2.shape[0])
    return y.permute(0, 3, 1, 2)                     # NCHW-shaped, channels-last-strided: _to_tokens takes it without a copy


def _resolve_dtype(x: Tensor) -> torch.dtype:
    if not x.is_cuda:
        raise RuntimeError("lemevit_amd: the model runs on an MI355X only -- move the model and inputs to 'cuda' "
                           "(the CPU restatement under oracle/ is test infrastructure, not a fallback)")
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        if dt != torch.bfloat16:
            raise NotImplementedError("lemevit_amd: autocast dtype must be torch.bfloat16 (fp16 kernels are not built)")
        return dt
    if x.dtype in (torch.float32, torch.bfloat16):
        return x.dtype
    raise NotImplementedError(f"lemevit_amd: unsupported input dtype {x.dtype}")


# ------------------------------------------------------------------------------------------------
# native block schedules (csrc/block.hip): ONE C-ABI call per block and pass instead of ~12 / ~30 per-op calls from blocks.py
# ------------------------------------------------------------------------------------------------
_NATIVE = os.environ.get("LMV_BLOCK_NATIVE", "1") != "0"      # 0: the Python schedules of blocks.py (A/B runs; they also serve "D2" / "Sx")
_KIND_CODE = {"S": 0, "D": 1, "C": 2}
_COMMON_FIELDS = {"pos_embed.weight": "pos_w", "pos_embed.bias": "pos_b", "norm1.weight": "n1_w", "norm1.bias": "n1_b", "norm2.weight": "n2_w",
                  "norm2.bias": "n2_b", "mlp.0.weight": "fc1_w", "mlp.0.bias": "fc1_b", "mlp.3.weight": "fc2_w", "mlp.3.bias": "fc2_b"}
_ATTN_SLOTS = {"S": {"attn.qkv": 0, "attn.proj": 1}, "D": {"attn.qkv1": 0, "attn.qkv2": 1, "attn.proj_x": 2, "attn.proj_c": 3},
               "C": {"attn.q": 0, "attn.kv": 1, "attn.proj": 2}}
_size_cache: dict = {}
_scratch_cache: dict = {}
_bwd_slot = 0


def _native_ok(kind: str, x: Tensor, c: Tensor) -> bool:
    return (_NATIVE and kind in _KIND_CODE and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and c.dtype == x.dtype
            and x.is_contiguous() and c.is_contiguous() and x.shape[-1] % 32 == 0)


def _fill_ptrs(desc, kind: str, names, tensors: Dict[str, Tensor], prefix: str) -> None:
    slots = _ATTN_SLOTS[kind]
    for n in names:
        t = tensors[n]
        if not t.is_contiguous() or not t.is_cuda:
            raise RuntimeError(f"lemevit_amd: block operand {n} must be a contiguous GPU tensor")
        f = _COMMON_FIELDS.get(n)
        if f is not None:
            setattr(desc, prefix + f, t.data_ptr())
        else:
            base, kindof = n.rsplit(".", 1)
            getattr(desc, prefix + ("attn_w" if kindof == "weight" else "attn_b"))[slots[base]] = t.data_ptr()


_WT_FIELDS = {"mlp.3.weight": ("fc2_wt", None), "mlp.0.weight": ("fc1_wt", None), "attn.qkv.weight": ("attn_wt", 0), "attn.proj.weight": ("attn_wt", 1)}


def _block_desc(kind: str, x: Tensor, c: Tensor, H: int, W: int, names, P: Dict[str, Tensor], masks, folds=None, wts: Optional[Dict[str, Tensor]] = None):
    from ._lib import BlockDesc
    d = BlockDesc()
    d.kind, d.dtype = _KIND_CODE[kind], ops.dtype_code(x)
    d.B, d.H, d.W, d.M, d.C = x.shape[0], H, W, c.shape[1], x.shape[2]
    d.hidden = P["mlp.0.weight"].shape[0]
    d.eps = BLOCK_LN_EPS
    for n in names:
        want = x.dtype if _is_matrix(n) else torch.float32
        if P[n].dtype != want:
            raise TypeError(f"lemevit_amd: block operand {n} must be {want}")
    _fill_ptrs(d, kind, names, P, "")
    for i, m in enumerate(masks):
        if m is not None:
            if m.dtype != torch.float32 or not m.is_contiguous():
                raise TypeError("lemevit_amd: DropPath scale vectors must be contiguous float32")
            d.masks[i] = m.data_ptr()
    if wts:                                 # transposed weight copies (FlatAdamW keeps them): those dX run as forward-form GEMMs
        d._wts = wts
        for n, wt in wts.items():
            w, (field, slot) = P[n], _WT_FIELDS[n]
            if wt.dtype == w.dtype == torch.bfloat16 and wt.shape == (w.shape[1], w.shape[0]) and wt.is_contiguous() and wt.device == w.device:
                if slot is None:
                    setattr(d, field, wt.data_ptr())
                else:
                    getattr(d, field)[slot] = wt.data_ptr()
    if folds is not None:                   # LMV_BLOCK_FUSED: inference with LayerNorm folded into the projections / the one-kernel MLP half
        attn, fc1 = folds
        for k, F in enumerate(attn):
            d.fold_attn_w[k], d.fold_attn_s[k], d.fold_attn_b[k] = F.wf.data_ptr(), F.colsum.data_ptr(), F.bf.data_ptr()
        d.fold_fc1_w, d.fold_fc1_s, d.fold_fc1_b = fc1.wf.data_ptr(), fc1.colsum.data_ptr(), fc1.bf.data_ptr()
        d.flags |= 2
        d._folds = folds                    # keep the operands alive as long as the descriptor
    return d


def _sized(fn, d, kind, x, c, H, W) -> int:
    key = (fn.__name__, kind, x.shape[0], H, W, c.shape[1], x.shape[2], x.dtype, d.hidden)
    n = _size_cache.get(key)
    if n is None:
        n = _size_cache[key] = int(fn(d))
        if n == 0:
            raise RuntimeError(f"{fn.__name__} failed: {ops.lib.lmv_last_error().decode(errors='replace')}")
    return n


def _persistent(tag: str, nbytes: int, device) -> Tensor:
    """Stream-ordered scratch that outlives the call (one per device, stream and use)."""
    key = (tag, device, ops._stream())
    t = _scratch_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = _scratch_cache[key] = torch.empty(int(nbytes * 1.1) + 4096, device=device, dtype=torch.uint8)
    return t


# ---- training forward as concurrent image ranges ------------------------------------------------------------------------------------------
# The images of a batch do not interact inside a LeMeBlock, so the blocks of one stage can run as TRAIN_PARTS independent ranges of images on
# forked streams (lmv_block_fwd_range: same arena, same outputs); the ramp and the tail of each launch of one range are then covered by the
# other ranges' kernels instead of an idle chip, as in lemevit_amd.graph.split_forward for inference.  The stage's downsample (a BatchNorm
# over the whole batch, models/lemevit.py:663-676) and the backward pass see whole-batch tensors: image_ranges() joins at the stage's end.
TRAIN_PARTS = int(os.environ.get("LMV_TRAIN_PARTS", "2"))
TRAIN_PARTS_MIN_BATCH = 32          # smaller batches are launch-bound: more, smaller launches do not pay
_range_state: dict = {}             # device index -> [streams, forked?]


class image_ranges:
    """with image_ranges(device, batch): the native block forwards inside (save=True) run as concurrent ranges of images."""

    def __init__(self, device, batch: int, parts: Optional[int] = None):
        parts = TRAIN_PARTS if parts is None else parts
        self.on = device.type == "cuda" and parts > 1 and batch >= max(TRAIN_PARTS_MIN_BATCH, parts) and torch.is_grad_enabled()
        self.dev, self.parts = device, parts

    def __enter__(self):
        if self.on:
            from . import blocks as _blocks
            streams = _blocks.aux_streams(self.dev, self.parts - 1)      # [0] = the weight-gradient side stream, idle during the forward pass
            _range_state[self.dev.index] = [streams, False, -1]
        return self

    def __exit__(self, *exc):
        if self.on:
            streams, forked, _ = _range_state.pop(self.dev.index)
            if forked:
                cur = torch.cuda.current_stream(self.dev)
                for s in streams:
                    cur.wait_stream(s)
        return False


def _join_ranges(device) -> None:
    """Anything inside image_ranges() that is NOT a range-aware native block (the per-launch Python schedule) reads whole-batch tensors on the
    current stream: the range streams are joined first."""
    rs = _range_state.get(device.index)
    if rs is not None and rs[1]:
        cur = torch.cuda.current_stream(device)
        for s in rs[0]:
            cur.wait_stream(s)
        rs[1] = False


def native_block_forward(kind: str, x: Tensor, c: Tensor, H: int, W: int, names, P: Dict[str, Tensor], masks, save: bool, folds=None, wts=None):
    """lmv_block_fwd: returns (x_out, c_out, state) with state = (descriptor, arena) for native_block_backward (save=True).
    Inside image_ranges() (training): one lmv_block_fwd_range call per range of images, each on its own stream."""
    from ._lib import lib, check
    d = _block_desc(kind, x, c, H, W, names, P, masks, None if save else folds, wts if save else None)
    nbytes = _sized(lib.lmv_block_arena_bytes, d, kind, x, c, H, W)
    arena = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if save else _persistent("fwd", nbytes, x.device)
    xo = torch.empty_like(x) if kind != "C" else None
    co = torch.empty_like(c)
    rs = _range_state.get(x.device.index) if save else None
    if rs is None:
        check(lib.lmv_block_fwd(d, x.data_ptr(), c.data_ptr(), None if xo is None else xo.data_ptr(), co.data_ptr(), arena.data_ptr(), arena.numel(), 1 if save else 0,
                                ops._stream()), "lmv_block_fwd")
    else:
        # The range streams wait for what the current stream holds at the FIRST block of the stage: the producers of x / c and whatever still
        # reads memory the allocator hands out again (nothing is freed inside the stage loop).  After that the ranges are only ordered within
        # their own stream (range i of block k+1 reads what range i of block k wrote on the same stream) -- unless a weight cast was launched
        # on the current stream since (parameters without an optimizer-maintained bf16 copy): then they wait for it.
        streams = rs[0]
        B, n = x.shape[0], len(rs[0]) + 1
        cuts = [B * i // n for i in range(n + 1)]
        cur = torch.cuda.current_stream(x.device)
        ev = None
        if not rs[1] or rs[2] != _casts_launched:
            ev = cur.record_event()
            rs[2] = _casts_launched
        for i in range(n):
            st = cur if i == 0 else streams[i - 1]
            if i and ev is not None:
                st.wait_event(ev)
            check(lib.lmv_block_fwd_range(d, x.data_ptr(), c.data_ptr(), None if xo is None else xo.data_ptr(), co.data_ptr(), arena.data_ptr(), arena.numel(), 1,
                                          cuts[i], cuts[i + 1] - cuts[i], st.cuda_stream), "lmv_block_fwd_range")
        rs[1] = True
    return (x if xo is None else xo), co, ((d, arena) if save else None)


def native_block_backward(kind: str, state, x: Tensor, c: Tensor, dx: Optional[Tensor], dc: Tensor, H: int, W: int, names, G: Dict[str, Tensor]):
    from ._lib import lib, check
    from . import blocks as blocks_mod
    from .blocks import side_stream_handle
    d, arena = state
    for n in names:
        if G[n].dtype != torch.float32:
            raise TypeError("lemevit_amd: gradient accumulators must be float32")
    _fill_ptrs(d, kind, names, G, "g_")
    nbytes = _sized(lib.lmv_block_bwd_scratch_bytes, d, kind, x, c, H, W)
    side = side_stream_handle(x.device)
    defer = side is not None and blocks_mod._DEFER > 0
    # deferred join (blocks.defer_join): the side stream may still read this block's scratch while the next blocks run -> rotate buffers
    global _bwd_slot
    _bwd_slot = (_bwd_slot + 1) % (blocks_mod._DEFER + 1) if defer else 0
    scratch = _persistent(("bwd", _bwd_slot), nbytes, x.device)
    dx0, dc0 = torch.empty_like(x), torch.empty_like(c)
    d.flags = 1 if defer else 0          # LMV_BLOCK_NO_JOIN
    check(lib.lmv_block_bwd(d, x.data_ptr(), c.data_ptr(), arena.data_ptr(), arena.numel(), None if dx is None else dx.data_ptr(), dc.data_ptr(), dx0.data_ptr(),
                            dc0.data_ptr(), scratch.data_ptr(), scratch.numel(), ops._stream(), side), "lmv_block_bwd")
    if defer:
        blocks_mod.defer_join(x.device.index, (arena, x, c, dx, dc, scratch))
    return dx0, dc0


# ------------------------------------------------------------------------------------------------
# one autograd node per block
# ------------------------------------------------------------------------------------------------
class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c, kind, H, W, masks, names, *params):
        cd = x.dtype
        P = {n: compute_copy(p, cd if _is_matrix(n) else torch.float32) for n, p in zip(names, params)}
        ctx.native = _native_ok(kind, x, c)
        if ctx.native:
            # a transposed copy is only valid next to the shadow it was made from (FlatAdamW refreshes both in one step)
            wts = {}
            for n in _WT_FIELDS:
                if n in names:
                    w = params[names.index(n)]
                    sh, wt = getattr(w, "_lmv_shadow", None), getattr(w, "_lmv_shadow_t", None)
                    if wt is not None and sh is not None and (P[n] is sh or P[n].data_ptr() == sh.data_ptr()):
                        wts[n] = wt
            xo, co, state = native_block_forward(kind, x, c, H, W, names, P, masks, save=True, wts=wts)
            saved = (x, c, state)
        else:
            if x.is_cuda:
                _join_ranges(x.device)
            xo, co, saved = block_forward(kind, x, c, H, W, P, masks, save=True)
        ctx.kind, ctx.H, ctx.W, ctx.masks, ctx.names = kind, H, W, masks, names
        ctx.saved, ctx.P = saved, P
        ctx.pmeta = [(p.shape, p.dtype) for p in params]
        ctx.params = params
        ctx.cshape = c.shape
        if kind == "C":
            return co                      # x passes through unchanged outside the node (models/lemevit.py:610)
        if kind == "Sx":
            return xo                      # c passes through unchanged outside the node (dense-prediction S block)
        return xo, co

    @staticmethod
    def backward(ctx, *grads):
        kind, names, P = ctx.kind, ctx.names, ctx.P
        if kind == "C":
            dx, dc = None, grads[0]
        elif kind == "Sx":
            dx, dc = grads[0], None
        else:
            dx, dc = grads
        x0 = ctx.saved[0]
        if dc is None and kind != "Sx":
            dc = torch.zeros(ctx.cshape, device=x0.device, dtype=x0.dtype)
        if dx is None and kind != "C":
            dx = torch.zeros_like(x0)
        # lemevit_amd.optim.FlatAdamW keeps every block parameter's .grad as a slice of one flat fp32 buffer: the kernels then
        # accumulate straight into it and autograd is handed None (no per-parameter accumulation launches)
        inplace = all(getattr(p, "_lmv_flat_grad", False) and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
                      for p in ctx.params)
        if inplace:
            G = {n: p.grad for n, p in zip(names, ctx.params)}
        else:
            sizes = [int(torch.Size(s).numel()) for s, _ in ctx.pmeta]
            pad = [(n + 3) // 4 * 4 for n in sizes]                 # keep every slice 16-byte aligned
            flat = torch.zeros(sum(pad), device=x0.device, dtype=torch.float32)
            G, off = {}, 0
            for n, (shape, _), sz, pd in zip(names, ctx.pmeta, sizes, pad):
                G[n] = flat[off:off + sz].view(shape)
                off += pd
        if ctx.native:
            xin, cin, state = ctx.saved
            dx0, dc0 = native_block_backward(kind, state, xin, cin, None if dx is None else dx.contiguous(), dc.contiguous(), ctx.H, ctx.W, names, G)
            if kind == "C" and dx is not None:
                dx0 = dx0 + dx                      # the untouched x's pass-through gradient (as blocks.block_backward)
        else:
            dx0, dc0 = block_backward(kind, ctx.saved, None if dx is None else dx.contiguous(), None if dc is None else dc.contiguous(), ctx.H, ctx.W, P, G, ctx.masks)
        cb = getattr(ctx.params[0], "_lmv_grad_cb", None) if inplace else None
        ctx.saved = ctx.params = None
        if cb is not None or not inplace:
            # the parameter gradients leave this node now (all-reduce of the chunk / autograd's accumulation on the current stream):
            # the deferred joins of the weight-gradient side stream (blocks.defer_join) are due
            from . import blocks as _blocks
            _blocks._wait_pending(0, x0.device.index)
        if cb is not None:
            cb()                     # lemevit_amd.dist.FlatGradSync: this block closes a chunk of the flat gradient buffer -> start its all-reduce
        if inplace:
            return (dx0, dc0, None, None, None, None, None, *([None] * len(names)))
        pg = [G[n] if dt == torch.float32 else G[n].to(dt) for n, (_, dt) in zip(names, ctx.pmeta)]
        return (dx0, dc0, None, None, None, None, None, *pg)


def run_block(kind: str, x: Tensor, c: Tensor, H: int, W: int, params: "OrderedDict[str, Tensor]",
              masks: Sequence[Optional[Tensor]]) -> Tuple[Tensor, Tensor]:
    """LeMeBlock on token-major tensors; picks the autograd node or the no-grad fast path."""
    names = PARAM_NAMES[kind]
    plist = [params[n] for n in names]
    need_grad = torch.is_grad_enabled() and (x.requires_grad or c.requires_grad or any(p.requires_grad for p in plist))
    if need_grad:
        out = _BlockFn.apply(x, c, kind, H, W, tuple(masks), names, *plist)
        return (x, out) if kind == "C" else ((out, c) if kind == "Sx" else out)
    cd = x.dtype
    P = {n: compute_copy(p, cd if _is_matrix(n) else torch.float32) for n, p in zip(names, plist)}
    folds = block_folds(kind, params, cd)
    if _native_ok(kind, x, c):
        xo, co, _ = native_block_forward(kind, x, c, H, W, names, P, masks, save=False, folds=folds)
        return xo, co
    xo, co, _ = block_forward(kind, x, c, H, W, P, masks, save=False, folds=folds)
    return xo, co


# ------------------------------------------------------------------------------------------------
# attention modules (parameter containers with the reference's names + inference forward)
# ------------------------------------------------------------------------------------------------
def _lin_probs(pairs, dtype):
    return [Prob(a, compute_copy(m.weight, dtype), o, bias=compute_copy(m.bias, torch.float32)) for a, m, o in pairs]


# ------------------------------------------------------------------------------------------------
# a whole stage of "S" blocks as one persistent launch (csrc/sstage.hip; inference)
# ------------------------------------------------------------------------------------------------
_SSTAGE = os.environ.get("LMV_SSTAGE", "1") != "0"        # 0: the per-block inference schedule (A/B runs)


class _LaunchContext(threading.local):
    """How many forward passes the CALLING host thread keeps in flight on the device at once (graph.split_forward sets it while it issues its sub-batches): the persistent stage
    kernels of a shape may only share the chip up to lmv_*stage_max_concurrent launches.  Thread-local: two host threads doing inference do not see each other's setting (each
    thread's own launches are what its streams overlap; threads that share a device on top of that are outside the per-thread bound, as processes are)."""
    concurrent = 1


launches = _LaunchContext()
_DSTAGE = os.environ.get("LMV_DSTAGE", "1") != "0"        # 0: stages of D blocks on the per-block schedule (A/B runs)
_INFER_SIDE = os.environ.get("LMV_INFER_SIDE", "1") != "0"        # inference: the meta-token MLP of a stage on a forked stream next to its transition convolution (0: in line)
_INFER_TAIL_PARTS = int(os.environ.get("LMV_INFER_TAIL_PARTS", "2"))      # inference: sub-batches of the per-launch stages behind the last persistent stage kernel (1: whole batch)
_sstage_cache: dict = {}


def _sstage_applies(stage, xt: Tensor, c: Tensor, H: int, W: int) -> Optional[str]:
    """A whole stage as ONE launch: models/lemevit.py:615-650 x depth (csrc/sstage.hip: every block an "S" block of a shape lmv_sstage_supported accepts: stage 3 of
    LeMeViT-Base / -Tiny at 224 x 224 -> "S") or :542-582 x depth (csrc/dstage.hip: "D" blocks, lmv_dstage_supported: stage 2 of LeMeViT-Base -> "D").  Inference only
    (nothing is saved for a backward pass, no DropPath), bf16.  None: the per-block schedule."""
    if not (_SSTAGE and _FUSED and _NATIVE) or ops.stage_kernels_disabled or torch.is_grad_enabled() or xt.dtype != torch.bfloat16 or len(stage) == 0 or not xt.is_cuda:
        return None
    kind = getattr(stage[0], "kind", None)
    if kind not in ("S", "D", "D2", "C"):
        return None
    for blk in stage:
        if type(blk) is not LeMeBlock or blk.kind != kind or (blk.training and blk.drop_prob > 0.0) or type(blk)._masks is not LeMeBlock._masks or "_masks" in blk.__dict__:
            return None
    b0 = stage[0]
    if kind == "S":
        if ops.sstage_supported(xt.shape[2], b0.attn.num_heads, b0.mlp[0].out_features, H, W, c.shape[1], xt.dtype):
            return "S" if launches.concurrent <= ops.sstage_max_concurrent(xt.shape[2]) else None
        # longer sequences (24 x 24 image tokens at 384 x 384): the multi-workgroup kernel of the D stages with self-attention across the workgroups of an image
        if _DSTAGE and ops.dstage_supported(xt.shape[2], b0.attn.num_heads, b0.mlp[0].out_features, H, W, c.shape[1], xt.dtype) and \
                launches.concurrent <= ops.dstage_max_concurrent(xt.shape[2], H, 2):
            return "S2"
        return None
    if not (_DSTAGE and ops.dstage_supported(xt.shape[2], b0.attn.num_heads, b0.mlp[0].out_features, H, W, c.shape[1], xt.dtype)):
        return None
    if launches.concurrent > ops.dstage_max_concurrent(xt.shape[2], H, 1 if kind == "C" else 0):
        return None          # graph.split_forward runs more sub-batches side by side than launches of this shape may share the chip (96 x 96 grids: one)
    if kind == "D":
        return "D" if all(type(blk.attn) is DualCrossAttention and blk.attn.scale == xt.shape[2] ** (-0.5) for blk in stage) else None
    if kind == "D2":          # (lemevit_tiny_v2: the same kernel, other packing -- ops.d2stage_pack)
        return "D2" if all(type(blk.attn) is DualCrossAttention_v2 and blk.attn.scale == xt.shape[2] ** (-0.5) for blk in stage) else None
    return "C" if all(type(blk.attn) is CrossAttention for blk in stage) else None          # stage 0: only the meta tokens change (:584-612)


def _whole_stage_fwd(whole: str, xt: Tensor, c: Tensor, packed, H: int, W: int):
    if whole == "S":
        return ops.sstage_fwd(xt, c, packed, H, W, BLOCK_LN_EPS, concurrent=launches.concurrent)
    return ops.dstage_fwd(xt, c, packed, H, W, BLOCK_LN_EPS, kind={"D": 0, "D2": 0, "C": 1, "S2": 2}[whole], concurrent=launches.concurrent)


def _sstage_packed(stage, kind: str = "S") -> "ops.SStagePacked":
    """The stage's parameters in the kernel's layout, cached per parameter version (and per training pass, see compute_copy)."""
    key = (id(stage), kind)          # one stage resolves to different kernels at different resolutions ("S" at 14 x 14, "S2" at 24 x 24): one pack per (stage, kind)
    plist = [p for blk in stage for p in blk._params().values()]
    stamp = tuple(p._version for p in plist) + tuple(id(p) for p in plist) + (_train_pass,)
    ent = _sstage_cache.get(key)
    if ent is not None and ent[0]() is stage and ent[1] == stamp:
        return ent[2]
    blocks = []
    for blk in stage:
        P = blk._params()
        d = {}
        for n in {"S": ops.SSTAGE_NAMES, "S2": ops.SSTAGE_NAMES, "D": ops.DSTAGE_NAMES, "D2": ops.D2STAGE_NAMES, "C": ops.CSTAGE_NAMES}[kind]:
            if n == "pos_embed.weight":
                d[n] = P[n].detach().float().reshape(P[n].shape[0], 9).contiguous()
            elif _is_matrix(n):
                d[n] = compute_copy(P[n], torch.bfloat16)
            else:
                d[n] = compute_copy(P[n], torch.float32)
        blocks.append(d)
    packed = {"S": ops.sstage_pack, "S2": ops.s2stage_pack, "D": ops.dstage_pack, "D2": ops.d2stage_pack, "C": ops.cstage_pack}[kind](blocks, stage[0].attn.num_heads)
    _cache_filled()
    _sstage_cache[key] = (weakref.ref(stage, lambda _r, k=key: _sstage_cache.pop(k, None)), stamp, packed)
    return packed


# ------------------------------------------------------------------------------------------------
# the attention modules as nn.Module sub-boundaries (DualCrossAttention.forward(x, c) -> (x, c) models/lemevit.py:252,
# StandardAttention.forward(x) -> x :185, CrossAttention.forward(x, c) -> c :454): called on their own they are ordinary differentiable modules --
# ONE autograd node over the same kernels the block schedules launch (projection GEMMs, attention cores with the log-sum-exp saved, their backward forms).
# ------------------------------------------------------------------------------------------------
def _needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


class _AttnModuleFn(torch.autograd.Function):
    """kind "S": (x) -> x'; "D" / "D2": (x, c) -> (x', c'); "C": (x, c) -> c'.  params: (weight, bias) of the module's Linears in declaration order."""

    @staticmethod
    def forward(ctx, kind, x, c, *params):
        new_training_pass()                      # parameters may have been updated in place since the last call (see compute_copy)
        cd = x.dtype
        x = x.contiguous()
        c = None if c is None else c.contiguous()
        W = [compute_copy(params[2 * i], cd) for i in range(len(params) // 2)]
        Bv = [compute_copy(params[2 * i + 1], torch.float32) for i in range(len(params) // 2)]
        C = x.shape[-1]
        emp = lambda like, cols: torch.empty(like.shape[:-1] + (cols,), device=like.device, dtype=cd)
        if kind == "S":
            qkv = emp(x, 3 * C)
            ops.linear_fwd([Prob(x, W[0], qkv, bias=Bv[0])], 3 * C, C)
            ao, lse = ops.attn_fwd((qkv, 0), (qkv, C), (qkv, 2 * C), C, ops.SDPA_SCALE, want_lse=True)
            out = torch.empty_like(x)
            ops.linear_fwd([Prob(ao, W[1], out, bias=Bv[1])], C, C)
            saved, outs = (x, qkv, ao, lse), (out,)
        elif kind in ("D", "D2"):
            N, M = x.shape[1], c.shape[1]
            sx, sc = ops.dca_scales(N, M, C)
            wd = 3 * C if kind == "D" else 2 * C
            p1, p2 = emp(x, wd), emp(c, wd)
            ops.linear_fwd([Prob(x, W[0], p1, bias=Bv[0]), Prob(c, W[1], p2, bias=Bv[1])], wd, C)
            if kind == "D":      # models/lemevit.py:297,300
                aox, lsex = ops.attn_fwd((p1, 0), (p2, C), (p2, 2 * C), C, sx, want_lse=True)
                aoc, lsec = ops.attn_fwd((p2, 0), (p1, C), (p1, 2 * C), C, sc, want_lse=True)
            else:                # models/lemevit.py:402,405: the same q / k serve both directions
                aox, lsex = ops.attn_fwd((p1, 0), (p2, 0), (p2, C), C, sx, want_lse=True)
                aoc, lsec = ops.attn_fwd((p2, 0), (p1, 0), (p1, C), C, sc, want_lse=True)
            ox, oc = torch.empty_like(x), torch.empty_like(c)
            ops.linear_fwd([Prob(aox, W[2], ox, bias=Bv[2]), Prob(aoc, W[3], oc, bias=Bv[3])], C, C)
            saved, outs = (x, c, p1, p2, aox, aoc, lsex, lsec), (ox, oc)
        else:                    # "C": q from the meta tokens, kv from the image tokens (models/lemevit.py:477-486)
            q, kv = torch.empty_like(c), emp(x, 2 * C)
            ops.linear_fwd([Prob(c, W[0], q, bias=Bv[0])], C, C)
            ops.linear_fwd([Prob(x, W[1], kv, bias=Bv[1])], 2 * C, C)
            ao, lse = ops.attn_fwd((q, 0), (kv, 0), (kv, C), C, ops.SDPA_SCALE, want_lse=True)
            out = torch.empty_like(c)
            ops.linear_fwd([Prob(ao, W[2], out, bias=Bv[2])], C, C)
            saved, outs = (x, c, q, kv, ao, lse), (out,)
        ctx.kind, ctx.W, ctx.acts, ctx.shapes = kind, W, saved, [(p.shape, p.dtype) for p in params]
        ctx.out_like = [(o.shape, o.dtype, o.device) for o in outs]
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *gs):
        kind, W, S = ctx.kind, ctx.W, ctx.acts
        # (an output the caller did not use arrives as None: its gradient is zero)
        gs = [torch.zeros(sh, dtype=dt, device=dv) if g is None else g.contiguous() for g, (sh, dt, dv) in zip(gs, ctx.out_like)]
        dev = gs[0].device
        G = [torch.zeros(s, device=dev, dtype=torch.float32) for s, _ in ctx.shapes]      # fp32 accumulators: (weight, bias) pairs
        dw = lambda probs, N, K: ops.linear_dw(probs, N, K)
        if kind == "S":
            x, qkv, ao, lse = S
            C = x.shape[-1]
            dw([Prob(gs[0], ao, G[2], bias_grad=G[3])], C, C)
            dao = torch.empty_like(x)
            ops.linear_dx([Prob(gs[0], W[1], dao)], C, C)
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd((qkv, 0), (qkv, C), (qkv, 2 * C), ao, lse, dao, (dqkv, 0), (dqkv, C), (dqkv, 2 * C), C, ops.SDPA_SCALE)
            dw([Prob(dqkv, x, G[0], bias_grad=G[1])], 3 * C, C)
            dx = torch.empty_like(x)
            ops.linear_dx([Prob(dqkv, W[0], dx)], 3 * C, C)
            dc = None
        elif kind in ("D", "D2"):
            x, c, p1, p2, aox, aoc, lsex, lsec = S
            C, N, M = x.shape[-1], x.shape[1], c.shape[1]
            sx, sc = ops.dca_scales(N, M, C)
            wd = p1.shape[-1]
            dw([Prob(gs[0], aox, G[4], bias_grad=G[5]), Prob(gs[1], aoc, G[6], bias_grad=G[7])], C, C)
            daox, daoc = torch.empty_like(x), torch.empty_like(c)
            ops.linear_dx([Prob(gs[0], W[2], daox), Prob(gs[1], W[3], daoc)], C, C)
            d1, d2 = torch.empty_like(p1), torch.empty_like(p2)
            if kind == "D":
                ops.attn_bwd((p1, 0), (p2, C), (p2, 2 * C), aox, lsex, daox, (d1, 0), (d2, C), (d2, 2 * C), C, sx)
                ops.attn_bwd((p2, 0), (p1, C), (p1, 2 * C), aoc, lsec, daoc, (d2, 0), (d1, C), (d1, 2 * C), C, sc)
            else:                # k is the query of the second direction, q its key: their gradients add to the first direction's
                ops.attn_bwd((p1, 0), (p2, 0), (p2, C), aox, lsex, daox, (d1, 0), (d2, 0), (d2, C), C, sx)
                t2, t1 = torch.empty_like(p2), torch.empty_like(p1)
                ops.attn_bwd((p2, 0), (p1, 0), (p1, C), aoc, lsec, daoc, (t2, 0), (t1, 0), (d1, C), C, sc)
                d1[..., :C] += t1[..., :C]
                d2[..., :C] += t2[..., :C]
            dw([Prob(d1, x, G[0], bias_grad=G[1]), Prob(d2, c, G[2], bias_grad=G[3])], wd, C)
            dx, dc = torch.empty_like(x), torch.empty_like(c)
            ops.linear_dx([Prob(d1, W[0], dx), Prob(d2, W[1], dc)], wd, C)
        else:
            x, c, q, kv, ao, lse = S
            C = x.shape[-1]
            dw([Prob(gs[0], ao, G[4], bias_grad=G[5])], C, C)
            dao = torch.empty_like(c)
            ops.linear_dx([Prob(gs[0], W[2], dao)], C, C)
            dq, dkv = torch.empty_like(q), torch.empty_like(kv)
            ops.attn_bwd((q, 0), (kv, 0), (kv, C), ao, lse, dao, (dq, 0), (dkv, 0), (dkv, C), C, ops.SDPA_SCALE)
            dw([Prob(dq, c, G[0], bias_grad=G[1])], C, C)
            dw([Prob(dkv, x, G[2], bias_grad=G[3])], 2 * C, C)
            dc, dx = torch.empty_like(c), torch.empty_like(x)
            ops.linear_dx([Prob(dq, W[0], dc)], C, C)
            ops.linear_dx([Prob(dkv, W[1], dx)], 2 * C, C)
        grads = [g.to(dt) for g, (_, dt) in zip(G, ctx.shapes)]
        return (None, dx, dc) + tuple(grads)


def _attn_params(*linears):
    out = []
    for m in linears:
        out += [m.weight, m.bias]
    return out


class StandardAttention(nn.Module):
    """models/lemevit.py:156-217.  forward(x [B,L,C]) -> [B,L,C] (inference path; training runs inside the block node)."""

    def __init__(self, dim, num_heads, scale=None, bias=False, attn_drop=0.0, proj_drop=0.0, **kwargs):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} not divisible by num_heads {num_heads}"
        assert dim // num_heads == ops.HEAD_DIM, "lemevit_amd kernels are built for head_dim 32 (all registered variants)"
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        P = _attn_params(self.qkv, self.proj)
        if _needs_grad(x, *P):
            return _AttnModuleFn.apply("S", x, None, *P)
        with torch.no_grad():
            return self._forward_nograd(x)

    def _forward_nograd(self, x):
        x = x.contiguous()
        C = x.shape[-1]
        qkv = torch.empty(x.shape[:-1] + (3 * C,), device=x.device, dtype=x.dtype)
        ops.linear_fwd(_lin_probs([(x, self.qkv, qkv)], x.dtype), 3 * C, C)
        ao, _ = ops.attn_fwd((qkv, 0), (qkv, C), (qkv, 2 * C), C, ops.SDPA_SCALE)
        out = torch.empty_like(x)
        ops.linear_fwd(_lin_probs([(ao, self.proj, out)], x.dtype), C, C)
        return out


class DualCrossAttention(nn.Module):
    """models/lemevit.py:220-324.  forward(x [B,N,C], c [B,M,C]) -> (x', c')."""

    def __init__(self, dim, num_heads, scale=None, bias=False, attn_drop=0.0, proj_drop=0.0, **kwargs):
        super().__init__()
        assert dim % num_heads == 0 and dim // num_heads == ops.HEAD_DIM
        self.num_heads = num_heads
        self.scale = scale or dim ** (-0.5)
        self.qkv1 = nn.Linear(dim, 3 * dim)
        self.qkv2 = nn.Linear(dim, 3 * dim)
        self.proj_x = nn.Linear(dim, dim)
        self.proj_c = nn.Linear(dim, dim)

    def forward(self, x, c):
        P = _attn_params(self.qkv1, self.qkv2, self.proj_x, self.proj_c)
        if _needs_grad(x, c, *P):
            return _AttnModuleFn.apply("D", x, c, *P)
        with torch.no_grad():
            return self._forward_nograd(x, c)

    def _forward_nograd(self, x, c):
        x, c = x.contiguous(), c.contiguous()
        C, N, M = x.shape[-1], x.shape[1], c.shape[1]
        sx, sc = ops.dca_scales(N, M, C)
        q1 = torch.empty(x.shape[:-1] + (3 * C,), device=x.device, dtype=x.dtype)
        q2 = torch.empty(c.shape[:-1] + (3 * C,), device=x.device, dtype=x.dtype)
        ops.linear_fwd(_lin_probs([(x, self.qkv1, q1), (c, self.qkv2, q2)], x.dtype), 3 * C, C)
        aox, _ = ops.attn_fwd((q1, 0), (q2, C), (q2, 2 * C), C, sx)
        aoc, _ = ops.attn_fwd((q2, 0), (q1, C), (q1, 2 * C), C, sc)
        ox, oc = torch.empty_like(x), torch.empty_like(c)
        ops.linear_fwd(_lin_probs([(aox, self.proj_x, ox), (aoc, self.proj_c, oc)], x.dtype), C, C)
        return ox, oc


class DualCrossAttention_v2(nn.Module):
    """models/lemevit.py:326-423 ("D2", lemevit_tiny_v2): q, v1 = qv1(x); k, v2 = kv2(c); x' = sdpa(q,k,v2), c' = sdpa(k,q,v1)."""

    def __init__(self, dim, num_heads, scale=None, bias=False, attn_drop=0.0, proj_drop=0.0, **kwargs):
        super().__init__()
        assert dim % num_heads == 0 and dim // num_heads == ops.HEAD_DIM
        self.num_heads = num_heads
        self.scale = scale or dim ** (-0.5)
        self.qv1 = nn.Linear(dim, 2 * dim)
        self.kv2 = nn.Linear(dim, 2 * dim)
        self.proj_x = nn.Linear(dim, dim)
        self.proj_c = nn.Linear(dim, dim)

    def forward(self, x, c):
        P = _attn_params(self.qv1, self.kv2, self.proj_x, self.proj_c)
        if _needs_grad(x, c, *P):
            return _AttnModuleFn.apply("D2", x, c, *P)
        with torch.no_grad():
            return self._forward_nograd(x, c)

    def _forward_nograd(self, x, c):
        x, c = x.contiguous(), c.contiguous()
        C, N, M = x.shape[-1], x.shape[1], c.shape[1]
        sx, sc = ops.dca_scales(N, M, C)
        qv1 = torch.empty(x.shape[:-1] + (2 * C,), device=x.device, dtype=x.dtype)
        kv2 = torch.empty(c.shape[:-1] + (2 * C,), device=x.device, dtype=x.dtype)
        ops.linear_fwd(_lin_probs([(x, self.qv1, qv1), (c, self.kv2, kv2)], x.dtype), 2 * C, C)
        aox, _ = ops.attn_fwd((qv1, 0), (kv2, 0), (kv2, C), C, sx)
        aoc, _ = ops.attn_fwd((kv2, 0), (qv1, 0), (qv1, C), C, sc)
        ox, oc = torch.empty_like(x), torch.empty_like(c)
        ops.linear_fwd(_lin_probs([(aox, self.proj_x, ox), (aoc, self.proj_c, oc)], x.dtype), C, C)
        return ox, oc


class CrossAttention(nn.Module):
    """models/lemevit.py:425-497.  forward(x [B,N,C], c [B,M,C]) -> c'."""

    def __init__(self, dim, num_heads, scale=None, bias=False, attn_drop=0.0, proj_drop=0.0, **kwargs):
        super().__init__()
        assert dim % num_heads == 0 and dim // num_heads == ops.HEAD_DIM
        self.num_heads = num_heads
        self.q = nn.Linear(dim, dim)
        self.kv = nn.Linear(dim, 2 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, c):
        P = _attn_params(self.q, self.kv, self.proj)
        if _needs_grad(x, c, *P):
            return _AttnModuleFn.apply("C", x, c, *P)
        with torch.no_grad():
            return self._forward_nograd(x, c)

    def _forward_nograd(self, x, c):
        x, c = x.contiguous(), c.contiguous()
        C = x.shape[-1]
        kv = torch.empty(x.shape[:-1] + (2 * C,), device=x.device, dtype=x.dtype)
        q = torch.empty_like(c)
        ops.linear_fwd(_lin_probs([(x, self.kv, kv)], x.dtype), 2 * C, C)
        ops.linear_fwd(_lin_probs([(c, self.q, q)], x.dtype), C, C)
        ao, _ = ops.attn_fwd((q, 0), (kv, 0), (kv, C), C, ops.SDPA_SCALE)
        out = torch.empty_like(c)
        ops.linear_fwd(_lin_probs([(ao, self.proj, out)], x.dtype), C, C)
        return out


class LeMeBlock(nn.Module):
    """models/lemevit.py:500-660 (pre_norm=True, no layer scale, cpe_ks=3, mlp_dwconv=False: the shipped variants)."""

    def __init__(self, dim, attn_drop, proj_drop, drop_path=0.0, attn_type=None, layer_scale_init_value=-1, num_heads=8, qk_dim=None,
                 mlp_ratio=4, mlp_dwconv=False, cpe_ks=3, pre_norm=True, dense=False):
        super().__init__()
        self.dense = bool(dense)            # dense-prediction backbones: an "S" block leaves the meta tokens untouched
        if layer_scale_init_value > 0 or not pre_norm or mlp_dwconv or cpe_ks != 3:
            raise NotImplementedError("lemevit_amd builds the live path of the shipped variants: pre_norm, no layer scale, cpe_ks=3, no mlp_dwconv")
        self.pos_embed = nn.Conv2d(dim, dim, kernel_size=cpe_ks, padding=1, groups=dim)
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn_type = attn_type or "S"
        if self.attn_type == "D":
            self.attn = DualCrossAttention(dim=dim, num_heads=num_heads)
        elif self.attn_type == "D2":
            self.attn = DualCrossAttention_v2(dim=dim, num_heads=num_heads)
        elif self.attn_type == "S":
            self.attn = StandardAttention(dim=dim, num_heads=num_heads)
        elif self.attn_type == "C":
            self.attn = CrossAttention(dim=dim, num_heads=num_heads)
        else:
            raise NotImplementedError(f"attention type {attn_type!r}")
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(dim, int(mlp_ratio * dim)), nn.Identity(), nn.GELU(), nn.Linear(int(mlp_ratio * dim), dim))
        self.drop_prob = float(drop_path)
        self._pcache = None

    def _params(self) -> "OrderedDict[str, Tensor]":
        if self._pcache is None:
            table = dict(self.named_parameters())
            self._pcache = OrderedDict((n, table[n]) for n in PARAM_NAMES[self.attn_type])
        return self._pcache

    def _apply(self, fn, *a, **k):     # .to()/.cuda() may replace Parameters
        self._pcache = None
        return super()._apply(fn, *a, **k)

    @property
    def kind(self) -> str:
        return "Sx" if (self.dense and self.attn_type == "S") else self.attn_type

    def _masks(self, B: int, device) -> List[Optional[Tensor]]:
        n = 2 if self.kind in ("C", "Sx") else 4
        if not self.training or self.drop_prob <= 0.0:
            return [None] * 4
        keep = 1.0 - self.drop_prob            # timm DropPath: per-sample Bernoulli(keep) / keep, drawn independently per call
        m = [torch.empty(B, device=device, dtype=torch.float32).bernoulli_(keep).div_(keep) for _ in range(n)]
        return m + [None] * (4 - n)

    def forward_tokens(self, x: Tensor, c: Tensor, H: int, W: int, masks=None) -> Tuple[Tensor, Tensor]:
        """x [B, H*W, C] token-major, c [B, M, C]."""
        if masks is None:
            masks = self._masks(x.shape[0], x.device)
            if any(m is not None for m in masks):
                _cache_filled()          # masks drawn HERE are kernels on the current stream: image_ranges() must re-fork its range streams behind them
        return run_block(self.kind, x, c, H, W, self._params(), masks)

    def forward(self, x: Tensor, c: Tensor) -> Tuple[Tensor, Tensor]:
        """Reference signature: x NCHW in / out (models/lemevit.py:652)."""
        if torch.is_grad_enabled():
            new_training_pass()        # parameters may have been updated in place since the last call (see compute_copy)
        B, C, H, W = x.shape
        xt = x.permute(0, 2, 3, 1).reshape(B, H * W, C).contiguous()
        xt, c = self.forward_tokens(xt, c.contiguous(), H, W)
        return xt.reshape(B, H, W, C).permute(0, 3, 1, 2), c


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class LeMeViT(nn.Module):
    """models/lemevit.py:663-836 -- same constructor, attributes, methods and state_dict layout."""

    def __init__(self, depth=[2, 3, 4, 8, 3], in_chans=3, num_classes=1000, embed_dim=[64, 64, 128, 320, 512], head_dim=64,
                 mlp_ratios=[4, 4, 4, 4, 4], qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop=0.0, drop_path_rate=0.0,
                 attn_type=["C", "D", "D", "S", "S"], queries_len=128, qk_dims=None, cpe_ks=3, pre_norm=True, mlp_dwconv=False,
                 representation_size=None, layer_scale_init_value=-1, use_checkpoint_stages=[], dense_blocks=False):
        super().__init__()
        if representation_size:
            raise NotImplementedError("representation_size is unused by every registered variant")
        if head_dim != ops.HEAD_DIM:
            raise NotImplementedError("lemevit_amd kernels are built for head_dim 32 (all registered variants)")
        if queries_len > 16:
            raise NotImplementedError("lemevit_amd kernels are built for <= 16 meta tokens (all registered variants use 16)")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        qk_dims = qk_dims or embed_dim
        self.num_stages = len(attn_type)
        self.attn_type = list(attn_type)
        self.depth = list(depth)

        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(
            nn.Conv2d(in_chans, embed_dim[0] // 2, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1)), nn.BatchNorm2d(embed_dim[0] // 2), nn.GELU(),
            nn.Conv2d(embed_dim[0] // 2, embed_dim[0], kernel_size=(3, 3), stride=(2, 2), padding=(1, 1)), nn.BatchNorm2d(embed_dim[0])))
        for i in range(self.num_stages - 1):
            if attn_type[i] == "C":
                self.downsample_layers.append(nn.Identity())
            else:
                self.downsample_layers.append(nn.Sequential(
                    nn.Conv2d(embed_dim[i], embed_dim[i + 1], kernel_size=(3, 3), stride=(2, 2), padding=(1, 1)), nn.BatchNorm2d(embed_dim[i + 1])))

        self.queries_len = queries_len
        self.meta_tokens = nn.Parameter(torch.randn(self.queries_len, embed_dim[0]), requires_grad=True)
        self.meta_token_downsample = nn.ModuleList()
        for i in range(self.num_stages):
            cin = embed_dim[0] if i == 0 else embed_dim[i - 1]
            self.meta_token_downsample.append(nn.Sequential(
                nn.Linear(cin, cin * 4), nn.LayerNorm(cin * 4), nn.GELU(), nn.Linear(cin * 4, embed_dim[i]), nn.LayerNorm(embed_dim[i])))

        self.stages = nn.ModuleList()
        nheads = [dim // head_dim for dim in qk_dims]
        dp_rates = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depth))]
        cur = 0
        for i in range(self.num_stages):
            self.stages.append(nn.ModuleList([
                LeMeBlock(dim=embed_dim[i], attn_drop=attn_drop, proj_drop=drop_rate, drop_path=dp_rates[cur + j], attn_type=attn_type[i],
                          layer_scale_init_value=layer_scale_init_value, num_heads=nheads[i], qk_dim=qk_dims[i], mlp_ratio=mlp_ratios[i],
                          mlp_dwconv=mlp_dwconv, cpe_ks=cpe_ks, pre_norm=pre_norm, dense=dense_blocks) for j in range(depth[i])]))
            cur += depth[i]

        self.norm = nn.BatchNorm2d(embed_dim[-1])
        self.norm_c = nn.LayerNorm(embed_dim[-1])
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(embed_dim[-1], num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)
        self.default_cfg = _cfg()

    def _init_weights(self, m):                      # models/lemevit.py:789-796
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=""):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim[-1], num_classes) if num_classes > 0 else nn.Identity()

    # ---- boundary glue (PyTorch-ROCm library ops) --------------------------
    @staticmethod
    def _to_tokens(x: Tensor, dtype) -> Tuple[Tensor, int, int]:
        B, C, H, W = x.shape
        return x.permute(0, 2, 3, 1).reshape(B, H * W, C).to(dtype).contiguous(), H, W

    @staticmethod
    def _to_nchw(xt: Tensor, H: int, W: int) -> Tensor:
        B, N, C = xt.shape
        return xt.view(B, H, W, C).permute(0, 3, 1, 2)          # channels_last-strided view, no copy

    def _run_downsample(self, seq: nn.Module, x: Tensor) -> Tensor:
        """Conv-BN(-GELU-Conv-BN) of the stem / a stage transition (models/lemevit.py:713-728).  In inference (eval mode,
        no grad) every BatchNorm is folded into the weights of the convolution before it: y = conv(x; w*s, (b-mu)*s+beta),
        s = gamma / sqrt(var + eps) -- the same affine map, one kernel and one pass over the feature map less per layer."""
        if not isinstance(seq, nn.Sequential):
            return seq(x)
        cd = x.dtype if not torch.is_autocast_enabled() else torch.get_autocast_dtype("cuda")
        fold = not (self.training or torch.is_grad_enabled())
        mods, i = list(seq), 0
        if fold and _stem_applies(mods, x, cd):
            return _stem_fused(mods, x, cd)
        while i < len(mods):
            m = mods[i]
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) and mods[i + 1].track_running_stats else None
            if fold and isinstance(m, nn.Conv2d) and bn is not None:
                w, b, b32 = _folded_conv_bn(m, bn, cd)
                if _is_stem_conv1(m, x):
                    gelu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.GELU) and getattr(mods[i + 2], "approximate", "none") == "none"
                    x = _StemConv1Fn.apply(x, w, b32, cd, gelu)
                    i += 3 if gelu else 2
                elif _is_conv3x3s2(m, x) and cd in (torch.float32, torch.bfloat16):
                    x = _Conv3x3s2Fn.apply(x, w, b32, cd)
                    i += 2
                else:
                    # (no vendor-library convolution behind the native ones: the stock PyTorch-ROCm column lives in tools/stock_eager.py)
                    raise NotImplementedError(f"lemevit_amd: no native kernel for {m} on a {tuple(x.shape)} {x.dtype} map (the library carries the 3 x 3 / stride-2 / padding-1 "
                                              "convolutions of the LeMeViT stem and stage transitions, channels a multiple of 8, fp32 / bf16)")
            elif _is_stem_conv1(m, x) and cd in (torch.float32, torch.bfloat16):
                x = _StemConv1Fn.apply(x, m.weight, m.bias, cd)
                i += 1
            elif _is_conv3x3s2(m, x) and cd in (torch.float32, torch.bfloat16):
                x = _Conv3x3s2Fn.apply(x, m.weight, m.bias, cd)
                i += 1
            elif _bn_native(m, x):
                gelu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU) and getattr(mods[i + 1], "approximate", "none") == "none"
                x = _bn_train(m, x, gelu)
                i += 2 if gelu else 1
            else:
                if isinstance(m, nn.Conv2d):
                    raise NotImplementedError(f"lemevit_amd: no native kernel for {m} on a {tuple(x.shape)} {x.dtype} map (see above)")
                x = m(x)          # (element-wise glue of a caller-built Sequential: activation, Identity, an eval-mode normalisation under autograd)
                i += 1
        return x

    def _draw_drop_path(self, B: int, device) -> Dict[int, List[Optional[Tensor]]]:
        """All DropPath masks of one forward pass from ONE uniform draw (instead of ~120 tiny bernoulli_/div_ launches):
        per block and per DropPath call an independent per-sample Bernoulli(keep) / keep vector, as timm's DropPath
        (models/lemevit.py:531,561-564; block 0 has rate 0 -> identity)."""
        if not self.training:
            return {}
        plan = [(blk, 2 if blk.kind in ("C", "Sx") else 4) for st in self.stages for blk in st if blk.drop_prob > 0.0 and type(blk)._masks is LeMeBlock._masks
                and "_masks" not in blk.__dict__]
        if not plan:
            return {}
        nmask = sum(n for _, n in plan)
        keep = getattr(self, "_dp_keep", None)      # cached on the device: no per-step host-to-device copy
        if keep is None or keep.shape[0] != nmask or keep.device != torch.device(device):
            keep = torch.tensor([1.0 - blk.drop_prob for blk, n in plan for _ in range(n)], dtype=torch.float32, device=device)
            self._dp_keep = keep
        u = torch.rand((keep.shape[0], B), device=device, dtype=torch.float32)
        scale = (u < keep[:, None]).to(torch.float32) / keep[:, None]
        out, r = {}, 0
        for blk, n in plan:
            out[id(blk)] = [scale[r + q] for q in range(n)] + [None] * (4 - n)
            r += n
        return out

    def forward_features(self, x: Tensor, c: Optional[Tensor] = None, head=False) -> Tensor:
        """models/lemevit.py:809-829.  c = None hoists the batch-invariant meta-token prefix.
        head (internal): the classifier module to fuse into the tail node (forward() passes self.head); False = features only."""
        cd = _resolve_dtype(x)
        self._tail_done = False
        B = x.shape[0]
        if torch.is_grad_enabled():
            new_training_pass()
        hoist = c is None
        if hoist:
            c = self.meta_tokens.unsqueeze(0)
        infer = not (self.training or torch.is_grad_enabled())
        st = {"cd": cd, "masks": self._draw_drop_path(B, x.device), "head": head, "checked": False, "infer": infer,
              # inference concurrency inside ONE pass (round 6): the meta-token MLP of a stage next to its transition convolution, and the per-launch stages behind the last
              # persistent stage kernel as sub-batches on forked streams.  Off inside graph.split_forward (its sub-batches already fill the streams a process holds).
              "side": infer and _INFER_SIDE and launches.concurrent == 1 and x.is_cuda,
              "tail_parts": _INFER_TAIL_PARTS if (infer and launches.concurrent == 1 and x.is_cuda and B >= 32) else 1,
              "key": (tuple(x.shape[1:]), cd), "whole": [False] * self.num_stages}
        out = self._stages_from(0, x, None, 0, 0, c, hoist, st)
        if infer and launches.concurrent == 1:
            self.__dict__.setdefault("_whole_seen", {})[st["key"]] = tuple(st["whole"])          # which stages ran as persistent launches at this input shape: where the next pass may split
        return out

    def _meta_tokens_for(self, i: int, c: Tensor, hoist: bool, B: int, cd: torch.dtype) -> Tensor:
        """The meta tokens entering stage i: meta_token_downsample[i] (models/lemevit.py:731-743, :812, :819)."""
        mlp = self.meta_token_downsample[i]
        pre = None
        if hoist and not torch.is_grad_enabled() and not self.training:
            # inference: the first meta-token MLP sees the learned meta tokens only (models/lemevit.py:731-743, :812, :833) -- a constant of the weights, cached per parameter version
            plist = [self.meta_tokens] + list(mlp.parameters())
            stamp = (_train_pass, cd) + tuple(p._version for p in plist) + tuple(p.data_ptr() for p in plist)
            ent = getattr(self, "_meta0_cache", None)
            if ent is not None and ent[0] == stamp:
                pre = ent[1]
        if pre is not None:
            c = pre
        elif _is_meta_mlp(mlp, c, cd):
            l1, n1, _, l2, n2 = mlp
            c = _MetaMLPFn.apply(c, l1.weight, l1.bias, n1.weight, n1.bias, l2.weight, l2.bias, n2.weight, n2.bias, n1.eps, n2.eps, cd)
        else:
            c = mlp(c)
        if pre is None and hoist and not torch.is_grad_enabled() and not self.training:
            self._meta0_cache = (stamp, c.detach())
            _cache_filled()
        if hoist:
            c = c.expand(B, -1, -1)
        return c.to(cd).contiguous()

    def _stages_from(self, i0: int, x: Optional[Tensor], xt: Optional[Tensor], H: int, W: int, c: Tensor, hoist: bool, st: dict) -> Tensor:
        """Stages i0 .. end and the classifier tail.  x: the NCHW input image (i0 = 0); xt: the token-major map leaving stage i0 - 1."""
        cd, head = st["cd"], st["head"]
        B = x.shape[0] if xt is None else xt.shape[0]
        for i in range(i0, self.num_stages):
            down = i == 0 or not isinstance(self.downsample_layers[i], nn.Identity)
            if st["tail_parts"] > 1 and i > i0 and xt is not None:
                seen = self.__dict__.get("_whole_seen", {}).get(st["key"])
                if seen is not None and seen[i - 1] and not any(seen[i:]):
                    return self._tail_split(i, xt, H, W, c, st)
            cur = torch.cuda.current_stream(c.device) if st["side"] else None
            if st["side"] and down and i > 0 and not hoist:
                # the meta tokens of stage i only need the meta tokens of stage i - 1: their MLP (four small launches) runs on a forked stream next to the transition convolution
                from .blocks import aux_streams
                side = aux_streams(c.device, 1)[0]
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    c_in, c = c, self._meta_tokens_for(i, c, False, B, cd)
                    c_in.record_stream(side)
                x = self._run_downsample(self.downsample_layers[i], self._to_nchw(xt, H, W))
                xt, H, W = self._to_tokens(x, cd)
                cur.wait_stream(side)
                c.record_stream(cur)
            else:
                if down:
                    x = self._run_downsample(self.downsample_layers[i], x if xt is None else self._to_nchw(xt, H, W))
                    xt, H, W = self._to_tokens(x, cd)
                c = self._meta_tokens_for(i, c, hoist, B, cd)
            hoist = False
            whole = _sstage_applies(self.stages[i], xt, c, H, W)
            if whole is not None:
                st["whole"][i] = True
                if not st["checked"]:          # the verdict on the stage launches of EARLIER calls (pinned host word: no synchronisation): a lost hand-off raises here, late but never silently
                    ops.check_stage_errors("found by LeMeViT.forward_features; raised by an earlier call", sync=False)
                    st["checked"] = True
                xt, c = _whole_stage_fwd(whole, xt.contiguous(), c, _sstage_packed(self.stages[i], whole), H, W)
                continue
            all_masks = st["masks"]
            with image_ranges(xt.device, B):
                for blk in self.stages[i]:
                    xt, c = blk.forward_tokens(xt, c, H, W, masks=all_masks.get(id(blk)) if all_masks else None)
        return self._tail(xt, H, W, c, st)

    def _tail_split(self, i: int, xt: Tensor, H: int, W: int, c: Tensor, st: dict) -> Tensor:
        """Inference: stages i .. end run per launch (no persistent stage kernel takes them: stage 4 of every variant) -- short launches that under-fill the chip, each paying its ramp
        and its tail.  The images of a batch do not interact in eval mode, so the rest of the pass runs as sub-batches on forked streams (graph.split_forward's argument, applied only
        where it pays: behind the last persistent launch)."""
        from .blocks import aux_streams
        parts = max(1, min(st["tail_parts"], xt.shape[0]))
        cur = torch.cuda.current_stream(xt.device)
        xs, cs = xt.chunk(parts), c.chunk(parts)
        streams = aux_streams(xt.device, len(xs) - 1)
        sub = dict(st, tail_parts=1, side=False)
        fork = cur.record_event()
        fills = cache_fills()
        ys = [self._stages_from(i, None, xs[0].contiguous(), H, W, cs[0].contiguous(), False, sub)]
        cold = cache_fills() != fills          # sub-batch 0 filled an operand cache on the current stream: the others wait for all of it (graph.split_forward)
        for k, s in enumerate(streams):
            if cold:
                s.wait_stream(cur)
            else:
                s.wait_event(fork)
            with torch.cuda.stream(s):
                ys.append(self._stages_from(i, None, xs[k + 1].contiguous(), H, W, cs[k + 1].contiguous(), False, sub))
            xt.record_stream(s); c.record_stream(s)
        for s, y in zip(streams, ys[1:]):
            cur.wait_stream(s)
            y.record_stream(cur)
        return torch.cat(ys)

    def _tail(self, xt: Tensor, H: int, W: int, c: Tensor, st: dict) -> Tensor:
        cd, head = st["cd"], st["head"]
        B = xt.shape[0]
        bn = self.norm
        if head is not False and isinstance(self.pre_logits, nn.Identity) and (self.training or torch.is_grad_enabled()) and _tail_native(self.norm_c, head, xt, c, cd):
            # training: final BatchNorm (native kernels) -> LayerNorm(c) + both mean-pools + add (+ classifier) as ONE autograd node
            xn = self._to_nchw(xt, H, W)
            xn = _bn_train(bn, xn) if _bn_native(bn, xn) else bn(xn)
            xb = xn.permute(0, 2, 3, 1).reshape(B, H * W, -1)           # token-major view of the channels-last map: no copy
            hw, hb = (None, None) if head is None else (head.weight, head.bias)
            self._tail_done = head is not None
            return _TailFn.apply(xb, c, self.norm_c.weight, self.norm_c.bias, float(self.norm_c.eps), hw, hb, cd)
        if (head is not False and head is not None and not bn.training and bn.track_running_stats and isinstance(self.pre_logits, nn.Identity)
                and not torch.is_grad_enabled() and _tail_native(self.norm_c, head, xt, c, cd) and bn.running_var is not None):
            # inference: LayerNorm(c) -> both mean-pools + add with the eval-mode BatchNorm folded into the pool -> classifier: three launches
            # of the library (lmv_layernorm_fwd, lmv_token_mean2_affine_fwd, lmv_linear_fwd) instead of ~20 ATen kernels + a hipBLASLt GEMM
            self._tail_done = True
            return _tail_infer(self.norm_c, bn, head, xt, c, cd)
        cn = self.pre_logits(self.norm_c(c))
        if not bn.training and bn.track_running_stats and isinstance(self.pre_logits, nn.Identity):
            # inference: BatchNorm with running statistics is affine per channel, so it commutes with the spatial mean --
            # pool the tokens first and normalise [B, C] instead of [B, C, H, W]  (models/lemevit.py:815, 825)
            pooled = xt.float().mean(dim=1)
            scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps) if bn.affine else torch.rsqrt(bn.running_var.float() + bn.eps)
            xm = (pooled - bn.running_mean.float()) * scale + (bn.bias.float() if bn.affine else 0.0)
            return xm.to(cn.dtype) + cn.mean(dim=1)
        xn = self._to_nchw(xt, H, W)
        xn = _bn_train(bn, xn) if _bn_native(bn, xn) else bn(xn)
        xn = self.pre_logits(xn)
        return xn.flatten(2).mean(-1) + cn.mean(dim=1)

    def forward(self, x: Tensor) -> Tensor:
        head = self.head if isinstance(self.head, nn.Linear) else None
        x = self.forward_features(x, None, head=head)
        return x if self._tail_done else self.head(x)


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channels of an NCHW map (timm.models.layers.LayerNorm2d): the dense-prediction backbone declares four
    of these (`extra_norms`) without calling them; kept so that its checkpoints load with strict=True."""

    def forward(self, x: Tensor) -> Tensor:
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class LeMeViTBackbone(LeMeViT):
    """The multi-scale backbone of the reference's detection / segmentation / change-detection folders
    (object_detection/mmdet/models/backbones/lemevit.py:660-877): same stem, meta-token path and stages as the classifier,
    but (a) the "S" blocks run attention and MLP on the image tokens only and pass the meta tokens through (:615-643), and
    (b) forward returns the NCHW feature maps after stages 1..4 (strides 4, 8, 16, 32) instead of logits (:798-818).
    train() keeps every BatchNorm2d / LayerNorm in eval mode and freezes `frozen_stages` as the reference does (:827-841);
    init_weights(path) loads a classification checkpoint with the 'backbone.' / 'module.' prefixes stripped (:844-877)."""

    def __init__(self, depth=[2, 3, 4, 8, 3], in_chans=3, num_classes=1000, embed_dim=[64, 64, 128, 320, 512], head_dim=64,
                 mlp_ratios=[4, 4, 4, 4, 4], qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop=0.0, drop_path_rate=0.0,
                 attn_type=["C", "D", "D", "S", "S"], queries_len=128, qk_dims=None, cpe_ks=3, pre_norm=True, mlp_dwconv=False,
                 representation_size=None, layer_scale_init_value=-1, use_checkpoint_stages=[], frozen_stages=[-1], pretrained=None):
        super().__init__(depth=depth, in_chans=in_chans, num_classes=0, embed_dim=embed_dim, head_dim=head_dim, mlp_ratios=mlp_ratios,
                         qkv_bias=qkv_bias, qk_scale=qk_scale, drop_rate=drop_rate, attn_drop=attn_drop, drop_path_rate=drop_path_rate,
                         attn_type=attn_type, queries_len=queries_len, qk_dims=qk_dims, cpe_ks=cpe_ks, pre_norm=pre_norm, mlp_dwconv=mlp_dwconv,
                         representation_size=representation_size, layer_scale_init_value=layer_scale_init_value,
                         use_checkpoint_stages=use_checkpoint_stages, dense_blocks=True)
        self.num_classes = num_classes
        self.frozen_stages = list(frozen_stages)
        self.extra_norms = nn.ModuleList([LayerNorm2d(embed_dim[i + 1]) for i in range(self.num_stages - 1)])
        del self.head                                   # the reference's backbone has no classifier
        if pretrained:
            self.init_weights(pretrained)

    def forward_features(self, x: Tensor, c: Optional[Tensor] = None) -> List[Tensor]:
        cd = _resolve_dtype(x)
        B = x.shape[0]
        if torch.is_grad_enabled():
            new_training_pass()
        if c is None:
            c = self.meta_tokens.repeat(B, 1, 1)
        xt, H, W, outs = None, 0, 0, []
        all_masks = self._draw_drop_path(B, x.device)
        for i in range(self.num_stages):
            if i == 0 or not isinstance(self.downsample_layers[i], nn.Identity):
                x = self._run_downsample(self.downsample_layers[i], x if xt is None else self._to_nchw(xt, H, W))
                xt, H, W = self._to_tokens(x, cd)
            mlp = self.meta_token_downsample[i]
            if _is_meta_mlp(mlp, c, cd):
                l1, n1, _, l2, n2 = mlp
                c = _MetaMLPFn.apply(c, l1.weight, l1.bias, n1.weight, n1.bias, l2.weight, l2.bias, n2.weight, n2.bias, n1.eps, n2.eps, cd)
            else:
                c = mlp(c)
            c = c.to(cd).contiguous()
            for blk in self.stages[i]:
                xt, c = blk.forward_tokens(xt, c, H, W, masks=all_masks.get(id(blk)) if all_masks else None)
            if i > 0:
                outs.append(self._to_nchw(xt, H, W))
        return outs

    def forward(self, x: Tensor) -> List[Tensor]:
        return self.forward_features(x, None)

    def _freeze_stages(self):
        for i in self.frozen_stages:
            if i >= 0:
                for p in self.stages[i].parameters():
                    p.requires_grad = False

    def train(self, mode: bool = True):
        self._freeze_stages()
        super().train(mode)
        for m in self.modules():                        # freeze_bn = True in the reference: normalisation layers stay in eval mode
            if isinstance(m, (nn.BatchNorm2d, nn.LayerNorm)):
                m.eval()
        return self

    def init_weights(self, pretrained: Optional[str] = None):
        if pretrained is not None:
            from .registry import load_checkpoint
            return load_checkpoint(self, pretrained, strict=False)
