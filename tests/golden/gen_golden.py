#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  The reference model file is imported
unmodified with its absent third-party imports (timm, fairscale) replaced by minimal stand-ins
(SURVEY.md Appendix A); nothing from the reference is copied into the repo -- the fixtures hold
expected *outputs* only, inputs and weights are regenerated from ``detfill.py``.

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from detfill import det_tensor, fill_state_dict, sample  # noqa: E402

REF = "/root/reference/models/lemevit.py"
REG = {}
ONLY = set(sys.argv[sys.argv.index("--only") + 1].split(",")) if "--only" in sys.argv else set()
DP_LOG = []          # DropPath masks in draw order (training fixtures)


def _import_reference():
    class DropPath(nn.Module):          # timm.models.layers.DropPath semantics
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob = drop_prob
            self.scale_by_keep = scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            m = x.new_empty(shape).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                m.div_(keep)
            DP_LOG.append(m.reshape(-1).clone())
            return x * m

    def mk(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    timm = mk("timm"); tm = mk("timm.models"); tl = mk("timm.models.layers")
    tv = mk("timm.models.vision_transformer")
    fs = mk("fairscale"); fn = mk("fairscale.nn"); fc = mk("fairscale.nn.checkpoint")
    timm.models = tm; tm.layers = tl; tm.vision_transformer = tv; fs.nn = fn; fn.checkpoint = fc
    tm.register_model = lambda f: (REG.__setitem__(f.__name__, f), f)[1]
    tl.DropPath = DropPath
    tl.to_2tuple = lambda x: (x, x)
    tl.trunc_normal_ = lambda t, std=1.0, mean=0.0, a=-2.0, b=2.0: nn.init.trunc_normal_(t, mean, std, a, b)
    tv._cfg = lambda url="", **kw: dict(url=url, num_classes=1000, input_size=(3, 224, 224), pool_size=None,
                                        crop_pct=0.9, interpolation="bicubic", fixed_input_size=True,
                                        mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                                        first_conv="patch_embed.proj", classifier="head", **kw)
    fc.checkpoint_wrapper = lambda m, *a, **k: m
    spec = importlib.util.spec_from_file_location("ref_lemevit", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


REF_DENSE = "/root/reference/object_detection/mmdet/models/backbones/lemevit.py"


def _import_dense_reference():
    """The detection backbone file, unmodified; its mmdet / mmcv imports (registry decorator, logger, checkpoint loader) and
    timm's LayerNorm2d are replaced by minimal stand-ins.  Call after _import_reference() (timm / fairscale stubs)."""
    def mk(name, pkg=False):
        m = types.ModuleType(name)
        if pkg:
            m.__path__ = []
        sys.modules[name] = m
        return m

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    class LayerNorm2d(nn.LayerNorm):
        def forward(self, x):
            return nn.functional.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)

    sys.modules["timm.models.layers"].LayerNorm2d = LayerNorm2d
    mk("mmdet", True); mk("mmdet.models", True); mk("mmdet.models.backbones", True)
    mk("mmdet.models.builder").BACKBONES = _Registry()
    mk("mmdet.utils").get_root_logger = lambda *a, **k: None
    mk("mmcv", True)
    mk("mmcv.runner")._load_checkpoint = lambda *a, **k: {}
    spec = importlib.util.spec_from_file_location("mmdet.models.backbones.lemevit", REF_DENSE)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["mmdet.models.backbones.lemevit"] = mod
    spec.loader.exec_module(mod)
    return mod


def _load(module: nn.Module, prefix: str, seed: int):
    spec = {prefix + k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = fill_state_dict(spec, seed)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return module


def _save(name: str, meta: dict, arrays: dict):
    arrays = {k: np.asarray(v) for k, v in arrays.items()}
    arrays["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB")


def gen_attention(ref):
    torch.manual_seed(0)
    for name, C, h, Hs, B in [("dca_96", 96, 3, 56, 2), ("dca_192", 192, 6, 28, 2), ("dca_odd", 64, 2, 35, 1)]:
        N = Hs * Hs
        m = _load(ref.DualCrossAttention(dim=C, num_heads=h).eval(), "attn.", 11)
        x = det_tensor((B, N, C), name + ".x", 1); c = det_tensor((B, 16, C), name + ".c", 1)
        with torch.no_grad():
            xo, co = m(x, c)
        _save(name, dict(kind="dca", C=C, h=h, N=N, M=16, B=B, seed=11),
              dict(x_out=sample(xo), c_out=co.numpy()))
    for name, C, h, L, B in [("sa_384_196", 384, 12, 196, 2), ("sa_384_16", 384, 12, 16, 2), ("sa_odd", 64, 2, 49, 3)]:
        m = _load(ref.StandardAttention(dim=C, num_heads=h).eval(), "attn.", 12)
        x = det_tensor((B, L, C), name + ".x", 1)
        with torch.no_grad():
            xo = m(x)
        _save(name, dict(kind="sa", C=C, h=h, L=L, B=B, seed=12), dict(x_out=sample(xo, 16384)))
    for name, C, h, Hs, B in [("ca_96", 96, 3, 56, 2), ("ca_odd", 64, 2, 35, 1)]:
        N = Hs * Hs
        m = _load(ref.CrossAttention(dim=C, num_heads=h).eval(), "attn.", 13)
        x = det_tensor((B, N, C), name + ".x", 1); c = det_tensor((B, 16, C), name + ".c", 1)
        with torch.no_grad():
            co = m(x, c)
        _save(name, dict(kind="ca", C=C, h=h, N=N, M=16, B=B, seed=13), dict(c_out=co.numpy()))
    # "D2" variant (lemevit_tiny_v2)
    m = _load(ref.DualCrossAttention_v2(dim=96, num_heads=3).eval(), "attn.", 14)
    x = det_tensor((2, 784, 96), "dca2.x", 1); c = det_tensor((2, 16, 96), "dca2.c", 1)
    with torch.no_grad():
        xo, co = m(x, c)
    _save("dca2_96", dict(kind="dca2", C=96, h=3, N=784, M=16, B=2, seed=14), dict(x_out=sample(xo), c_out=co.numpy()))


def _block(ref, t, C, h, dp=0.0):
    return ref.LeMeBlock(dim=C, attn_drop=0.0, proj_drop=0.0, drop_path=dp, attn_type=t,
                         layer_scale_init_value=-1, num_heads=h, mlp_ratio=4, mlp_dwconv=False, cpe_ks=3, pre_norm=True)


def gen_blocks(ref):
    for name, t, C, h, Hs, B in [("block_C", "C", 64, 2, 28, 2), ("block_D", "D", 96, 3, 28, 2), ("block_S", "S", 192, 6, 14, 2)]:
        m = _load(_block(ref, t, C, h).eval(), "blk.", 21)
        x = det_tensor((B, C, Hs, Hs), name + ".x", 2); c = det_tensor((B, 16, C), name + ".c", 2)
        with torch.no_grad():
            xo, co = m(x, c)
        _save(name, dict(kind="block", type=t, C=C, h=h, H=Hs, W=Hs, B=B, seed=21),
              dict(x_out=sample(xo, 16384), c_out=co.numpy()))
    # forward + backward (grads wrt inputs and every parameter; checks shared-weight accumulation)
    for name, t, C, h, Hs, B in [("blockgrad_D", "D", 64, 2, 12, 2), ("blockgrad_S", "S", 64, 2, 7, 2), ("blockgrad_C", "C", 64, 2, 12, 2)]:
        m = _load(_block(ref, t, C, h).eval(), "blk.", 22)
        x = det_tensor((B, C, Hs, Hs), name + ".x", 3).requires_grad_(True)
        c = det_tensor((B, 16, C), name + ".c", 3).requires_grad_(True)
        gx = det_tensor((B, C, Hs, Hs), name + ".gx", 3); gc = det_tensor((B, 16, C), name + ".gc", 3)
        xo, co = m(x, c)
        ((xo * gx).sum() + (co * gc).sum()).backward()
        arr = dict(x_out=xo.detach().numpy(), c_out=co.detach().numpy(), dx=x.grad.numpy(), dc=c.grad.numpy())
        for k, p in m.named_parameters():
            arr["grad." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        _save(name, dict(kind="blockgrad", type=t, C=C, h=h, H=Hs, W=Hs, B=B, seed=22), arr)


def gen_models(ref):
    for name, variant, res, B, nc in [("model_tiny_224", "lemevit_tiny", 224, 2, 1000), ("model_base_224", "lemevit_base", 224, 2, 1000),
                                      ("model_small_224", "lemevit_small", 224, 1, 51), ("model_tiny_384", "lemevit_tiny", 384, 1, 1000),
                                      ("model_tiny_v2_224", "lemevit_tiny_v2", 224, 1, 1000), ("model_small_v2_224", "lemevit_small_v2", 224, 1, 1000),
                                      ("model_vit_tiny_224", "vit_tiny", 224, 1, 1000),
                                      ("model_base_384", "lemevit_base", 384, 1, 1000)]:      # BASELINE config 5 (round 2)
        if ONLY and name not in ONLY:
            continue
        torch.manual_seed(0)
        m = REG[variant](num_classes=nc).eval()
        nparams = sum(p.numel() for p in m.parameters())
        _load(m, "", 31)
        img = det_tensor((B, 3, res, res), name + ".img", 4)
        inter = []
        with torch.no_grad():
            c = m.meta_tokens.repeat(B, 1, 1)
            x = img
            for i in range(m.num_stages):
                x = m.downsample_layers[i](x); c = m.meta_token_downsample[i](c)
                for blk in m.stages[i]:
                    x, c = blk(x, c)
                inter.append((x, c))
            logits = m(img)
        arr = dict(logits=logits.numpy())
        for i, (x, c) in enumerate(inter):
            arr[f"stage{i}.x"] = sample(x.flatten(2).transpose(1, 2))      # token-major order [B, HW, C]
            arr[f"stage{i}.c"] = c.numpy()
        _save(name, dict(kind="model", variant=variant, res=res, B=B, num_classes=nc, seed=31, nparams=nparams,
                         nkeys=len(m.state_dict())), arr)


def gen_train(ref):
    for name, dpr in [("train_tiny_96", 0.0), ("train_tiny_96_dp", 0.3)]:
        torch.manual_seed(0)
        m = REG["lemevit_tiny"](num_classes=10, drop_path_rate=dpr)
        _load(m, "", 41)
        m.train()
        DP_LOG.clear()
        img = det_tensor((4, 3, 96, 96), name + ".img", 5)
        target = torch.tensor([1, 7, 3, 3])
        logits = m(img)
        loss = nn.functional.cross_entropy(logits, target)
        loss.backward()
        arr = dict(logits=logits.detach().numpy(), loss=np.float32(loss.item()))
        names = []
        gn = []
        for k, p in m.named_parameters():
            names.append(k); gn.append(float(p.grad.norm()) if p.grad is not None else 0.0)
        arr["grad_norms"] = np.asarray(gn, dtype=np.float32)
        for k in ["meta_tokens", "head.weight", "stages.1.0.attn.qkv1.weight", "stages.3.2.mlp.0.weight", "stages.0.0.attn.kv.weight",
                  "stages.2.1.norm1.weight", "stages.4.1.pos_embed.weight", "downsample_layers.0.0.weight", "stages.3.0.attn.qkv.bias"]:
            arr["grad." + k] = dict(m.named_parameters())[k].grad.numpy()
        for k, v in m.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                arr["stat." + k] = v.numpy()
        if DP_LOG:
            arr["dp_masks"] = torch.stack(DP_LOG).numpy()
        _save(name, dict(kind="train", variant="lemevit_tiny", res=96, B=4, num_classes=10, seed=41, drop_path_rate=dpr,
                         target=target.tolist(), param_names=names), arr)


def gen_dense(refd):
    """Dense-prediction backbone: multi-scale outputs, "S" blocks that leave the meta tokens alone."""
    tiny = dict(depth=[1, 2, 2, 8, 2], embed_dim=[64, 64, 128, 192, 320], head_dim=32, mlp_ratios=[4, 4, 4, 4, 4],
                attn_type=["C", "D", "D", "S", "S"], queries_len=16)
    for name, Hh, Ww, B in [("dense_tiny_224", 224, 224, 1), ("dense_tiny_160x96", 160, 96, 2)]:
        torch.manual_seed(0)
        m = refd.LeMeViT(**tiny)
        m.eval()                    # (the reference's train() override returns None)
        _load(m, "", 51)
        img = det_tensor((B, 3, Hh, Ww), name + ".img", 6)
        with torch.no_grad():
            outs = m(img)
        arr = {f"out{i}": sample(o.flatten(2).transpose(1, 2), 8192) for i, o in enumerate(outs)}
        _save(name, dict(kind="dense", cfg=tiny, H=Hh, W=Ww, B=B, seed=51, nkeys=len(m.state_dict()),
                         shapes=[list(o.shape) for o in outs]), arr)
    # one S block of the dense file, forward + backward: c must come back untouched and receive only its pass-through gradient
    blk = refd.LeMeBlock(dim=64, attn_drop=0.0, proj_drop=0.0, drop_path=0.0, attn_type="S", layer_scale_init_value=-1, num_heads=2,
                         mlp_ratio=4, mlp_dwconv=False, cpe_ks=3, pre_norm=True).eval()
    _load(blk, "blk.", 52)
    name, B, C, Hs = "blockgrad_Sx", 2, 64, 7
    x = det_tensor((B, C, Hs, Hs), name + ".x", 3).requires_grad_(True)
    c = det_tensor((B, 16, C), name + ".c", 3).requires_grad_(True)
    gx = det_tensor((B, C, Hs, Hs), name + ".gx", 3); gc = det_tensor((B, 16, C), name + ".gc", 3)
    xo, co = blk(x, c)
    ((xo * gx).sum() + (co * gc).sum()).backward()
    arr = dict(x_out=xo.detach().numpy(), c_out=co.detach().numpy(), dx=x.grad.numpy(), dc=c.grad.numpy())
    for k, p in blk.named_parameters():
        arr["grad." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    _save(name, dict(kind="blockgrad", type="Sx", C=C, h=2, H=Hs, W=Hs, B=B, seed=52), arr)


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    ref = _import_reference()
    assert ref.has_torchfunc and not ref.has_flash_attn and not ref.has_xformers
    if ONLY:                       # --only model_base_384[,...]: add a model fixture without rewriting the others
        gen_models(ref)
        return
    if "--dense-only" in sys.argv:
        gen_dense(_import_dense_reference())
        return
    gen_attention(ref)
    gen_blocks(ref)
    gen_models(ref)
    gen_train(ref)
    gen_dense(_import_dense_reference())


if __name__ == "__main__":
    main()
