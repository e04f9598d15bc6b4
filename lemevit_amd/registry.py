"""Minimal model registry with timm's calling convention (timm is not a dependency).

``create_model('lemevit_base', pretrained=False, num_classes=..., drop_path_rate=..., **kw)`` behaves like
``timm.create_model`` for the arguments the reference's callers pass (benchmark.py:409-419, main.py:170-184,
validate.py:210-218): ``None``-valued kwargs are dropped, ``scriptable`` / ``exportable`` / ``checkpoint_path`` /
``pretrained_cfg*`` are consumed here, the rest reaches ``LeMeViT(...)``.  If timm IS importable the factories are
also registered there, so ``timm.create_model('lemevit_base')`` resolves to this implementation.
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch

from .model import LeMeViT, _cfg

_REGISTRY: Dict[str, Callable] = {}


def register_model(fn: Callable) -> Callable:
    _REGISTRY[fn.__name__] = fn
    try:  # pragma: no cover - timm is absent in the build image
        from timm.models import register_model as _timm_register
        _timm_register(fn)
    except Exception:
        pass
    return fn


def list_models() -> List[str]:
    return sorted(_REGISTRY)


def is_model(name: str) -> bool:
    return name in _REGISTRY


def create_model(model_name: str, pretrained=False, pretrained_cfg=None, pretrained_cfg_overlay=None, checkpoint_path: str = "",
                 scriptable=None, exportable=None, no_jit=None, **kwargs):
    if model_name not in _REGISTRY:
        raise RuntimeError(f"Unknown model ({model_name}); available: {list_models()}")
    if scriptable:
        raise NotImplementedError("lemevit_amd models call a C-ABI kernel library and are not TorchScript-able")
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    for consumed in ("global_pool", "drop_block_rate", "bn_momentum", "bn_eps"):   # passed by timm-style callers, unused by LeMeViT
        kwargs.pop(consumed, None)
    model = _REGISTRY[model_name](pretrained=pretrained, **kwargs)
    if checkpoint_path:
        load_checkpoint(model, checkpoint_path)
    return model


def load_checkpoint(model: torch.nn.Module, path: str, strict: bool = True):
    """Accepts the reference's formats: {"model": sd} (models/lemevit.py:869-870), timm's {"state_dict": sd} /
    {"state_dict_ema": sd}, or a bare state_dict; strips DDP / detector prefixes ('module.', 'backbone.')."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt
    for key in ("model", "state_dict_ema", "state_dict"):
        if isinstance(ckpt, dict) and key in ckpt and isinstance(ckpt[key], dict):
            sd = ckpt[key]
            break
    clean = {}
    for k, v in sd.items():
        for prefix in ("module.", "backbone."):
            if k.startswith(prefix):
                k = k[len(prefix):]
        clean[k] = v
    return model.load_state_dict(clean, strict=strict)


def _variant(pretrained, kwargs, **hp):
    kwargs.pop("pretrained_cfg", None); kwargs.pop("pretrained_cfg_overlay", None)
    model = LeMeViT(head_dim=32, queries_len=16, qkv_bias=True, qk_scale=None, attn_drop=0.0, qk_dims=None, cpe_ks=3, pre_norm=True,
                    mlp_dwconv=False, representation_size=None, layer_scale_init_value=-1, use_checkpoint_stages=[], **hp, **kwargs)
    model.default_cfg = _cfg()
    if pretrained:                      # the reference treats `pretrained` as a checkpoint PATH (models/lemevit.py:868-870)
        load_checkpoint(model, pretrained)
    return model


@register_model
def lemevit_tiny(pretrained=False, pretrained_cfg=None, pretrained_cfg_overlay=None, **kwargs):
    """models/lemevit.py:845-872."""
    return _variant(pretrained, kwargs, depth=[1, 2, 2, 8, 2], embed_dim=[64, 64, 128, 192, 320], mlp_ratios=[4, 4, 4, 4, 4],
                    attn_type=["C", "D", "D", "S", "S"])


@register_model
def lemevit_small(pretrained=False, pretrained_cfg=None, pretrained_cfg_overlay=None, **kwargs):
    """models/lemevit.py:875-902."""
    return _variant(pretrained, kwargs, depth=[1, 2, 2, 6, 2], embed_dim=[96, 96, 192, 320, 384], mlp_ratios=[4, 4, 4, 4, 4],
                    attn_type=["C", "D", "D", "S", "S"])


@register_model
def lemevit_base(pretrained=False, pretrained_cfg=None, pretrained_cfg_overlay=None, **kwargs):
    """models/lemevit.py:905-932."""
    return _variant(pretrained, kwargs, depth=[2, 4, 4, 18, 4], embed_dim=[96, 96, 192, 384, 512], mlp_ratios=[4, 4, 4, 4, 4],
                    attn_type=["C", "D", "D", "S", "S"])


@register_model
def lemevit_small_v2(pretrained=False, pretrained_cfg=None, pretrained_cfg_overlay=None, **kwargs):
    """models/lemevit.py:935-962."""
    return _variant(pretrained, kwargs, depth=[1, 2, 2, 8, 2], embed_dim=[64, 64, 128, 256, 512], mlp_ratios=[3, 3, 3, 3, 3],
                    attn_type=["C", "D", "D", "S", "S"])


@register_model
def lemevit_tiny_v2(pretrained=False, pretrained_cfg=None, pretrained_cfg_overlay=None, **kwargs):
    """models/lemevit.py:965-992 (shared-q/k "D2" dual cross attention)."""
    return _variant(pretrained, kwargs, depth=[2, 2, 2, 4, 2], embed_dim=[96, 96, 192, 320, 384], mlp_ratios=[4, 4, 4, 4, 4],
                    attn_type=["C", "D2", "D2", "S", "S"])


@register_model
def vit_tiny(pretrained=False, pretrained_cfg=None, pretrained_cfg_overlay=None, **kwargs):
    """models/lemevit.py:996-1023 (all-"S" baseline)."""
    return _variant(pretrained, kwargs, depth=[2, 2, 4, 2], embed_dim=[96, 192, 320, 384], mlp_ratios=[4, 4, 4, 4],
                    attn_type=["S", "S", "S", "S"])
