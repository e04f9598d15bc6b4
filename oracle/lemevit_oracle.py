"""CPU oracle for the LeMeViT backbone hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a plain-PyTorch (CPU, fp32 or fp64) *restatement* of the algorithm in the
reference's ``models/lemevit.py``.  It exists so that the hand-written HIP kernels under
``lemevit_amd/csrc`` can be checked against something that follows the reference line by
line.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the shipped package (``lemevit_amd``) never does and fails
loudly when its HIP library is missing.

Pinning: the reference ships no tests or golden vectors for this path, so
the oracle is pinned against outputs of the reference itself: ``tests/golden/gen_golden.py``
imports ``/root/reference/models/lemevit.py`` (with timm/fairscale stubbed) in the build
container, and ``tests/test_oracle_golden.py`` checks this file against the committed
fixtures (max-abs <= 2e-6 fp32 on every fixture).  Known-answer pins the reference does
publish (parameter counts, README.md:85-87) are checked too.

The oracle is *functional*: weights come in as a ``state_dict`` with exactly the
reference's key names, activations are token-major ``[B, L, C]`` inside.
Because it is written with differentiable torch ops, ``torch.autograd`` over it yields the
reference gradients used to check the HIP backward kernels.

Reference citations (``models/lemevit.py``):
  sdpa                  :54-63      softmax(q k^T * scale) v
  StandardAttention     :156-217    live branch :199-205
  DualCrossAttention    :220-324    scales :235,255-256 ; live branch :288-302
  CrossAttention        :425-497    live branch :477-486
  LeMeBlock             :500-660    C :584-613, D :542-582, S :615-650 (pre_norm, no layer scale)
  LeMeViT               :663-836    stem :698-704, downsample :714-717, meta MLP :729-745,
                                    tail :815-835
  factories             :845-1023
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------------------
# hyper-parameters of the registered variants (models/lemevit.py:848-865, 878-895, 908-925,
# 938-955, 968-985, 999-1016)
# ----------------------------------------------------------------------------------------
VARIANTS: Dict[str, dict] = {
    "lemevit_tiny": dict(depth=[1, 2, 2, 8, 2], embed_dim=[64, 64, 128, 192, 320], head_dim=32,
                         mlp_ratios=[4, 4, 4, 4, 4], attn_type=["C", "D", "D", "S", "S"], queries_len=16),
    "lemevit_small": dict(depth=[1, 2, 2, 6, 2], embed_dim=[96, 96, 192, 320, 384], head_dim=32,
                          mlp_ratios=[4, 4, 4, 4, 4], attn_type=["C", "D", "D", "S", "S"], queries_len=16),
    "lemevit_base": dict(depth=[2, 4, 4, 18, 4], embed_dim=[96, 96, 192, 384, 512], head_dim=32,
                         mlp_ratios=[4, 4, 4, 4, 4], attn_type=["C", "D", "D", "S", "S"], queries_len=16),
    "lemevit_small_v2": dict(depth=[1, 2, 2, 8, 2], embed_dim=[64, 64, 128, 256, 512], head_dim=32,
                             mlp_ratios=[3, 3, 3, 3, 3], attn_type=["C", "D", "D", "S", "S"], queries_len=16),
    "lemevit_tiny_v2": dict(depth=[2, 2, 2, 4, 2], embed_dim=[96, 96, 192, 320, 384], head_dim=32,
                            mlp_ratios=[4, 4, 4, 4, 4], attn_type=["C", "D2", "D2", "S", "S"], queries_len=16),
    "vit_tiny": dict(depth=[2, 2, 4, 2], embed_dim=[96, 192, 320, 384], head_dim=32,
                     mlp_ratios=[4, 4, 4, 4], attn_type=["S", "S", "S", "S"], queries_len=16),
}

BLOCK_LN_EPS = 1e-6   # models/lemevit.py:513,525
META_LN_EPS = 1e-5    # nn.LayerNorm default, models/lemevit.py:731-743,774
BN_EPS = 1e-5         # nn.BatchNorm2d default, models/lemevit.py:700,703,716,773
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """Row-wise LayerNorm over the last dim (biased variance), as nn.LayerNorm."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """Exact GELU (nn.GELU default, models/lemevit.py:528,701,732)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def sdpa(q: Tensor, k: Tensor, v: Tensor, scale: Optional[float] = None) -> Tensor:
    """models/lemevit.py:54-63 (same math as F.scaled_dot_product_attention). [B,h,L,d]."""
    d = q.shape[-1]
    s = scale if scale is not None else d ** (-0.5)
    attn = (q @ k.transpose(-1, -2)) * s
    attn = attn.softmax(dim=-1)
    return attn @ v


def split_heads(t: Tensor, parts: int, h: int) -> Tuple[Tensor, ...]:
    """'B L (x h d) -> x B h L d' (models/lemevit.py:201,290,292,481)."""
    B, L, XC = t.shape
    d = XC // (parts * h)
    t = t.reshape(B, L, parts, h, d).permute(2, 0, 3, 1, 4)
    return tuple(t[i] for i in range(parts))


def merge_heads(t: Tensor) -> Tensor:
    """'B h L d -> B L (h d)' (models/lemevit.py:204,298,301,485)."""
    B, h, L, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(B, L, h * d)


def dca_scales(N: int, M: int, C: int) -> Tuple[float, float]:
    """models/lemevit.py:235,255-256: scale uses the FULL embed dim, scale_x = log_N(M) * C^-1/2."""
    base = C ** (-0.5)
    return math.log(M, N) * base, math.log(N, N) * base


# ----------------------------------------------------------------------------------------
# attention flavours (token-major in/out)
# ----------------------------------------------------------------------------------------
def standard_attention(sd, p: str, x: Tensor, h: int) -> Tensor:
    """StandardAttention.forward, models/lemevit.py:199-205."""
    qkv = linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    q, k, v = split_heads(qkv, 3, h)
    o = merge_heads(sdpa(q, k, v))
    return linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def dual_cross_attention(sd, p: str, x: Tensor, c: Tensor, h: int) -> Tuple[Tensor, Tensor]:
    """DualCrossAttention.forward, models/lemevit.py:252-256,288-302."""
    B, N, C = x.shape
    M = c.shape[1]
    scale_x, scale_c = dca_scales(N, M, C)
    q1, k1, v1 = split_heads(linear(x, sd[p + "qkv1.weight"], sd[p + "qkv1.bias"]), 3, h)
    q2, k2, v2 = split_heads(linear(c, sd[p + "qkv2.weight"], sd[p + "qkv2.bias"]), 3, h)
    xo = merge_heads(sdpa(q1, k2, v2, scale_x))
    xo = linear(xo, sd[p + "proj_x.weight"], sd[p + "proj_x.bias"])
    co = merge_heads(sdpa(q2, k1, v1, scale_c))
    co = linear(co, sd[p + "proj_c.weight"], sd[p + "proj_c.bias"])
    return xo, co


def dual_cross_attention_v2(sd, p: str, x: Tensor, c: Tensor, h: int) -> Tuple[Tensor, Tensor]:
    """DualCrossAttention_v2.forward ("D2"), models/lemevit.py:357-361,393-407."""
    B, N, C = x.shape
    M = c.shape[1]
    scale_x, scale_c = dca_scales(N, M, C)
    q, v1 = split_heads(linear(x, sd[p + "qv1.weight"], sd[p + "qv1.bias"]), 2, h)
    k, v2 = split_heads(linear(c, sd[p + "kv2.weight"], sd[p + "kv2.bias"]), 2, h)
    xo = merge_heads(sdpa(q, k, v2, scale_x))
    xo = linear(xo, sd[p + "proj_x.weight"], sd[p + "proj_x.bias"])
    co = merge_heads(sdpa(k, q, v1, scale_c))
    co = linear(co, sd[p + "proj_c.weight"], sd[p + "proj_c.bias"])
    return xo, co


def cross_attention(sd, p: str, x: Tensor, c: Tensor, h: int) -> Tensor:
    """CrossAttention.forward, models/lemevit.py:477-486 (queries = meta tokens)."""
    q = linear(c, sd[p + "q.weight"], sd[p + "q.bias"])
    kv = linear(x, sd[p + "kv.weight"], sd[p + "kv.bias"])
    (q,) = split_heads(q, 1, h)
    k, v = split_heads(kv, 2, h)
    o = merge_heads(sdpa(q, k, v))
    return linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


# ----------------------------------------------------------------------------------------
# LeMeBlock (models/lemevit.py:500-660), x token-major [B, H*W, C]
# ----------------------------------------------------------------------------------------
def pos_embed_residual(sd, p: str, x: Tensor, H: int, W: int) -> Tensor:
    """x + dwconv3x3(x)  (models/lemevit.py:510,546); x is [B, H*W, C] token-major."""
    B, N, C = x.shape
    xi = x.transpose(1, 2).reshape(B, C, H, W)
    y = F.conv2d(xi, sd[p + "pos_embed.weight"], sd[p + "pos_embed.bias"], stride=1, padding=1, groups=C)
    return x + y.reshape(B, C, N).transpose(1, 2)


def mlp(sd, p: str, x: Tensor) -> Tensor:
    """nn.Sequential(Linear, Identity, GELU, Linear), models/lemevit.py:526-530."""
    hdn = gelu_erf(linear(x, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
    return linear(hdn, sd[p + "mlp.3.weight"], sd[p + "mlp.3.bias"])


def _dp(branch: Tensor, mask: Optional[Tensor]) -> Tensor:
    """timm DropPath: per-sample scale (0 or 1/keep), shape [B]; None = identity."""
    if mask is None:
        return branch
    return branch * mask.reshape(-1, *([1] * (branch.dim() - 1)))


def leme_block(sd, p: str, attn_type: str, x: Tensor, c: Tensor, H: int, W: int, h: int,
               dp_masks: Optional[Sequence[Optional[Tensor]]] = None) -> Tuple[Tensor, Tensor]:
    """LeMeBlock.forward (pre_norm=True, no layer scale).  ``dp_masks`` = up to four per-sample
    DropPath scale vectors in the order the reference draws them."""
    m = list(dp_masks) if dp_masks is not None else [None] * 4
    n1 = lambda t: layer_norm(t, sd[p + "norm1.weight"], sd[p + "norm1.bias"], BLOCK_LN_EPS)
    n2 = lambda t: layer_norm(t, sd[p + "norm2.weight"], sd[p + "norm2.bias"], BLOCK_LN_EPS)
    x_in = x
    x = pos_embed_residual(sd, p, x, H, W)
    if attn_type in ("D", "D2"):          # forward_with_xc :560-564
        fn = dual_cross_attention if attn_type == "D" else dual_cross_attention_v2
        ax, ac = fn(sd, p + "attn.", n1(x), n1(c), h)
        x = x + _dp(ax, m[0])
        x = x + _dp(mlp(sd, p, n2(x)), m[1])
        c = c + _dp(ac, m[2])
        c = c + _dp(mlp(sd, p, n2(c)), m[3])
        return x, c
    if attn_type == "S":                  # forward_with_x :632-635
        x = x + _dp(standard_attention(sd, p + "attn.", n1(x), h), m[0])
        x = x + _dp(mlp(sd, p, n2(x)), m[1])
        c = c + _dp(standard_attention(sd, p + "attn.", n1(c), h), m[2])
        c = c + _dp(mlp(sd, p, n2(c)), m[3])
        return x, c
    if attn_type == "Sx":                 # dense-prediction backbones: forward_with_x touches x only and returns c as it came
        x = x + _dp(standard_attention(sd, p + "attn.", n1(x), h), m[0])       # (object_detection/mmdet/models/backbones/lemevit.py:615-643)
        x = x + _dp(mlp(sd, p, n2(x)), m[1])
        return x, c
    if attn_type == "C":                  # forward_with_c :600-601, returns the ORIGINAL x :610
        c = c + _dp(cross_attention(sd, p + "attn.", n1(x), n1(c), h), m[0])
        c = c + _dp(mlp(sd, p, n2(c)), m[1])
        return x_in, c
    raise NotImplementedError(attn_type)


# ----------------------------------------------------------------------------------------
# whole model
# ----------------------------------------------------------------------------------------
def batch_norm(sd, p: str, x: Tensor, train: bool, new_stats: Optional[dict] = None) -> Tensor:
    """nn.BatchNorm2d on NCHW input; train=True uses batch statistics (biased var)."""
    w, b = sd[p + "weight"], sd[p + "bias"]
    if train:
        mu = x.mean(dim=(0, 2, 3))
        var = ((x - mu[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        if new_stats is not None:
            n = x.numel() / x.shape[1]
            new_stats[p + "running_mean"] = (1 - BN_MOMENTUM) * sd[p + "running_mean"] + BN_MOMENTUM * mu.detach()
            new_stats[p + "running_var"] = (1 - BN_MOMENTUM) * sd[p + "running_var"] + BN_MOMENTUM * var.detach() * n / (n - 1)
    else:
        mu, var = sd[p + "running_mean"], sd[p + "running_var"]
    xh = (x - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def meta_mlp(sd, p: str, c: Tensor) -> Tensor:
    """meta_token_downsample[i]: Linear, LN, GELU, Linear, LN (models/lemevit.py:729-745)."""
    c = linear(c, sd[p + "0.weight"], sd[p + "0.bias"])
    c = layer_norm(c, sd[p + "1.weight"], sd[p + "1.bias"], META_LN_EPS)
    c = gelu_erf(c)
    c = linear(c, sd[p + "3.weight"], sd[p + "3.bias"])
    return layer_norm(c, sd[p + "4.weight"], sd[p + "4.bias"], META_LN_EPS)


def to_tokens(x: Tensor) -> Tuple[Tensor, int, int]:
    B, C, H, W = x.shape
    return x.reshape(B, C, H * W).transpose(1, 2), H, W


def to_nchw(x: Tensor, H: int, W: int) -> Tensor:
    B, N, C = x.shape
    return x.transpose(1, 2).reshape(B, C, H, W)


def lemevit_forward(sd: Dict[str, Tensor], cfg: dict, img: Tensor, train: bool = False,
                    dp_masks: Optional[Dict[Tuple[int, int], Sequence[Optional[Tensor]]]] = None,
                    intermediates: Optional[List] = None, new_stats: Optional[dict] = None) -> Tensor:
    """LeMeViT.forward (models/lemevit.py:809-836).  img: [B,3,H,W] -> logits [B,num_classes].

    dp_masks[(stage, block)] = four per-sample DropPath scale vectors (training only).
    intermediates, if a list, receives (x_tokens, c) after every stage."""
    depth, dims, types = cfg["depth"], cfg["embed_dim"], cfg["attn_type"]
    heads = [d // cfg["head_dim"] for d in dims]
    B = img.shape[0]
    c = sd["meta_tokens"].unsqueeze(0).repeat(B, 1, 1)                      # :833
    x = img
    for i in range(len(types)):
        # ---- downsample_layers[i] (:698-717)
        if i == 0:
            x = F.conv2d(x, sd["downsample_layers.0.0.weight"], sd["downsample_layers.0.0.bias"], stride=2, padding=1)
            x = batch_norm(sd, "downsample_layers.0.1.", x, train, new_stats)
            x = gelu_erf(x)
            x = F.conv2d(x, sd["downsample_layers.0.3.weight"], sd["downsample_layers.0.3.bias"], stride=2, padding=1)
            x = batch_norm(sd, "downsample_layers.0.4.", x, train, new_stats)
        elif types[i - 1] != "C":
            x = F.conv2d(x, sd[f"downsample_layers.{i}.0.weight"], sd[f"downsample_layers.{i}.0.bias"], stride=2, padding=1)
            x = batch_norm(sd, f"downsample_layers.{i}.1.", x, train, new_stats)
        # ---- meta_token_downsample[i] (:812)
        c = meta_mlp(sd, f"meta_token_downsample.{i}.", c)
        xt, H, W = to_tokens(x)
        for j in range(depth[i]):
            masks = None if dp_masks is None else dp_masks.get((i, j))
            xt, c = leme_block(sd, f"stages.{i}.{j}.", types[i], xt, c, H, W, heads[i], masks)
        x = to_nchw(xt, H, W)
        if intermediates is not None:
            intermediates.append((xt, c))
    x = batch_norm(sd, "norm.", x, train, new_stats)                         # :815
    c = layer_norm(c, sd["norm_c.weight"], sd["norm_c.bias"], META_LN_EPS)   # :818
    feat = x.flatten(2).mean(-1) + c.mean(dim=1)                             # :825-827
    return linear(feat, sd["head.weight"], sd["head.bias"])                  # :835


def lemevit_dense_forward(sd: Dict[str, Tensor], cfg: dict, img: Tensor, train_bn: bool = False,
                          dp_masks: Optional[Dict[Tuple[int, int], Sequence[Optional[Tensor]]]] = None) -> List[Tensor]:
    """The multi-scale backbone of the detection / segmentation folders
    (object_detection/mmdet/models/backbones/lemevit.py:798-824): the classifier's stem, meta-token path and stages with the
    "S" blocks leaving the meta tokens untouched (:615-643); returns the NCHW maps after stages 1..4."""
    depth, dims, types = cfg["depth"], cfg["embed_dim"], cfg["attn_type"]
    heads = [d // cfg["head_dim"] for d in dims]
    B = img.shape[0]
    c = sd["meta_tokens"].unsqueeze(0).repeat(B, 1, 1)
    x, outs = img, []
    for i in range(len(types)):
        if i == 0:
            x = F.conv2d(x, sd["downsample_layers.0.0.weight"], sd["downsample_layers.0.0.bias"], stride=2, padding=1)
            x = batch_norm(sd, "downsample_layers.0.1.", x, train_bn)
            x = gelu_erf(x)
            x = F.conv2d(x, sd["downsample_layers.0.3.weight"], sd["downsample_layers.0.3.bias"], stride=2, padding=1)
            x = batch_norm(sd, "downsample_layers.0.4.", x, train_bn)
        elif types[i - 1] != "C":
            x = F.conv2d(x, sd[f"downsample_layers.{i}.0.weight"], sd[f"downsample_layers.{i}.0.bias"], stride=2, padding=1)
            x = batch_norm(sd, f"downsample_layers.{i}.1.", x, train_bn)
        c = meta_mlp(sd, f"meta_token_downsample.{i}.", c)
        xt, H, W = to_tokens(x)
        for j in range(depth[i]):
            masks = None if dp_masks is None else dp_masks.get((i, j))
            xt, c = leme_block(sd, f"stages.{i}.{j}.", "Sx" if types[i] == "S" else types[i], xt, c, H, W, heads[i], masks)
        x = to_nchw(xt, H, W)
        if i > 0:
            outs.append(x)
    return outs


# ----------------------------------------------------------------------------------------
# state_dict construction (layout of SURVEY.md section 8b) and reference-style init (:726, :789-796)
# ----------------------------------------------------------------------------------------
def state_dict_spec(cfg: dict, num_classes: int = 1000, in_chans: int = 3) -> "Dict[str, Tuple[int, ...]]":
    """Ordered {key: shape} identical to the reference model's state_dict()."""
    depth, dims, types, ratios = cfg["depth"], cfg["embed_dim"], cfg["attn_type"], cfg["mlp_ratios"]
    spec: Dict[str, Tuple[int, ...]] = {}

    def lin(p, o, i):
        spec[p + ".weight"] = (o, i); spec[p + ".bias"] = (o,)

    def ln(p, n):
        spec[p + ".weight"] = (n,); spec[p + ".bias"] = (n,)

    def bn(p, n):
        ln(p, n); spec[p + ".running_mean"] = (n,); spec[p + ".running_var"] = (n,)
        spec[p + ".num_batches_tracked"] = ()

    def conv(p, o, i, k=3):
        spec[p + ".weight"] = (o, i, k, k); spec[p + ".bias"] = (o,)

    spec["meta_tokens"] = (cfg["queries_len"], dims[0])
    conv("downsample_layers.0.0", dims[0] // 2, in_chans); bn("downsample_layers.0.1", dims[0] // 2)
    conv("downsample_layers.0.3", dims[0], dims[0] // 2); bn("downsample_layers.0.4", dims[0])
    for i in range(1, len(types)):
        if types[i - 1] != "C":
            conv(f"downsample_layers.{i}.0", dims[i], dims[i - 1]); bn(f"downsample_layers.{i}.1", dims[i])
    for i in range(len(types)):
        cin = dims[0] if i == 0 else dims[i - 1]
        p = f"meta_token_downsample.{i}"
        lin(p + ".0", cin * 4, cin); ln(p + ".1", cin * 4); lin(p + ".3", dims[i], cin * 4); ln(p + ".4", dims[i])
    for i in range(len(types)):
        C = dims[i]
        for j in range(depth[i]):
            p = f"stages.{i}.{j}"
            spec[p + ".pos_embed.weight"] = (C, 1, 3, 3); spec[p + ".pos_embed.bias"] = (C,)
            ln(p + ".norm1", C)
            t = types[i]
            if t == "D":
                lin(p + ".attn.qkv1", 3 * C, C); lin(p + ".attn.qkv2", 3 * C, C)
                lin(p + ".attn.proj_x", C, C); lin(p + ".attn.proj_c", C, C)
            elif t == "D2":
                lin(p + ".attn.qv1", 2 * C, C); lin(p + ".attn.kv2", 2 * C, C)
                lin(p + ".attn.proj_x", C, C); lin(p + ".attn.proj_c", C, C)
            elif t == "S":
                lin(p + ".attn.qkv", 3 * C, C); lin(p + ".attn.proj", C, C)
            elif t == "C":
                lin(p + ".attn.q", C, C); lin(p + ".attn.kv", 2 * C, C); lin(p + ".attn.proj", C, C)
            ln(p + ".norm2", C)
            hid = int(ratios[i] * C)
            lin(p + ".mlp.0", hid, C); lin(p + ".mlp.3", C, hid)
    bn("norm", dims[-1]); ln("norm_c", dims[-1])
    if num_classes > 0:
        lin("head", num_classes, dims[-1])
    return spec


def count_params(cfg: dict, num_classes: int = 1000) -> int:
    """Number of trainable parameters (README.md:85-87 known answers)."""
    n = 0
    for k, shp in state_dict_spec(cfg, num_classes).items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        n += int(math.prod(shp)) if shp else 1
    return n
