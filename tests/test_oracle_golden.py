"""Pins oracle/lemevit_oracle.py against golden vectors produced by the reference itself
(tests/golden/gen_golden.py) and against the reference's published known answers."""
import math

import numpy as np
import pytest
import torch

from detfill import det_tensor, fill_state_dict, sample
from oracle import lemevit_oracle as O

TOL = 3e-6   # fp32 CPU restatement vs fp32 reference, max-abs on O(1) outputs


def _close(a, b, tol=TOL, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    ref = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * ref, f"{what}: max-abs {err:.3e} (ref max {ref:.3e})"


def _attn_sd(kind, C, seed):
    shapes = {"dca": {"qkv1": 3, "qkv2": 3, "proj_x": 1, "proj_c": 1}, "dca2": {"qv1": 2, "kv2": 2, "proj_x": 1, "proj_c": 1},
              "sa": {"qkv": 3, "proj": 1}, "ca": {"q": 1, "kv": 2, "proj": 1}}[kind]
    spec = {}
    for n, mult in shapes.items():
        spec[f"attn.{n}.weight"] = (mult * C, C); spec[f"attn.{n}.bias"] = (mult * C,)
    return fill_state_dict(spec, seed)


@pytest.mark.parametrize("name", ["dca_96", "dca_192", "dca_odd", "dca2_96"])
def test_dual_cross_attention(golden, name):
    meta, g = golden(name)
    C, h, N, B = meta["C"], meta["h"], meta["N"], meta["B"]
    sd = _attn_sd(meta["kind"], C, meta["seed"])
    pre = "dca2" if meta["kind"] == "dca2" else name
    x = det_tensor((B, N, C), pre + ".x", 1); c = det_tensor((B, 16, C), pre + ".c", 1)
    fn = O.dual_cross_attention if meta["kind"] == "dca" else O.dual_cross_attention_v2
    xo, co = fn(sd, "attn.", x, c, h)
    _close(sample(xo), g["x_out"], what=name + ".x"); _close(co, g["c_out"], what=name + ".c")


@pytest.mark.parametrize("name", ["sa_384_196", "sa_384_16", "sa_odd"])
def test_standard_attention(golden, name):
    meta, g = golden(name)
    sd = _attn_sd("sa", meta["C"], meta["seed"])
    x = det_tensor((meta["B"], meta["L"], meta["C"]), name + ".x", 1)
    _close(sample(O.standard_attention(sd, "attn.", x, meta["h"]), 16384), g["x_out"], what=name)


@pytest.mark.parametrize("name", ["ca_96", "ca_odd"])
def test_cross_attention(golden, name):
    meta, g = golden(name)
    sd = _attn_sd("ca", meta["C"], meta["seed"])
    x = det_tensor((meta["B"], meta["N"], meta["C"]), name + ".x", 1); c = det_tensor((meta["B"], 16, meta["C"]), name + ".c", 1)
    _close(O.cross_attention(sd, "attn.", x, c, meta["h"]), g["c_out"], what=name)


def block_spec(t, C, prefix="blk."):
    cfg = dict(depth=[1], embed_dim=[C], attn_type=[t], mlp_ratios=[4], queries_len=16, head_dim=32)
    full = O.state_dict_spec(cfg, num_classes=0)
    return {prefix + k[len("stages.0.0."):]: v for k, v in full.items() if k.startswith("stages.0.0.")}


@pytest.mark.parametrize("name", ["block_C", "block_D", "block_S"])
def test_block_forward(golden, name):
    meta, g = golden(name)
    t, C, h, H, W, B = meta["type"], meta["C"], meta["h"], meta["H"], meta["W"], meta["B"]
    sd = fill_state_dict(block_spec(t, C), meta["seed"])
    x = det_tensor((B, C, H, W), name + ".x", 2); c = det_tensor((B, 16, C), name + ".c", 2)
    xt, _, _ = O.to_tokens(x)
    xo, co = O.leme_block(sd, "blk.", t, xt, c, H, W, h)
    _close(sample(O.to_nchw(xo, H, W), 16384), g["x_out"], what=name + ".x"); _close(co, g["c_out"], what=name + ".c")


@pytest.mark.parametrize("name", ["blockgrad_D", "blockgrad_S", "blockgrad_C"])
def test_block_backward(golden, name):
    meta, g = golden(name)
    t, C, h, H, W, B = meta["type"], meta["C"], meta["h"], meta["H"], meta["W"], meta["B"]
    sd = {k: v.requires_grad_(True) for k, v in fill_state_dict(block_spec(t, C), meta["seed"]).items()}
    x = det_tensor((B, C, H, W), name + ".x", 3).requires_grad_(True); c = det_tensor((B, 16, C), name + ".c", 3).requires_grad_(True)
    gx = det_tensor((B, C, H, W), name + ".gx", 3); gc = det_tensor((B, 16, C), name + ".gc", 3)
    xt, _, _ = O.to_tokens(x)
    xo, co = O.leme_block(sd, "blk.", t, xt, c, H, W, h)
    xo = O.to_nchw(xo, H, W)
    ((xo * gx).sum() + (co * gc).sum()).backward()
    _close(xo.detach(), g["x_out"], what="x_out"); _close(co.detach(), g["c_out"], what="c_out")
    _close(x.grad, g["dx"], 1e-5, "dx"); _close(c.grad, g["dc"], 1e-5, "dc")
    for k, v in sd.items():
        gr = v.grad if v.grad is not None else torch.zeros_like(v)
        _close(gr, g["grad." + k[len("blk."):]], 2e-5, "grad " + k)


@pytest.mark.parametrize("name", ["model_tiny_224", "model_base_224", "model_small_224", "model_tiny_384", "model_base_384",
                                  "model_tiny_v2_224", "model_small_v2_224", "model_vit_tiny_224"])
def test_model_forward(golden, name):
    meta, g = golden(name)
    cfg = O.VARIANTS[meta["variant"]]
    spec = O.state_dict_spec(cfg, meta["num_classes"])
    assert len(spec) == meta["nkeys"]
    assert O.count_params(cfg, meta["num_classes"]) == meta["nparams"]
    sd = fill_state_dict(spec, meta["seed"])
    img = det_tensor((meta["B"], 3, meta["res"], meta["res"]), name + ".img", 4)
    inter = []
    with torch.no_grad():
        logits = O.lemevit_forward(sd, cfg, img, intermediates=inter)
    for i, (x, c) in enumerate(inter):
        _close(sample(x), g[f"stage{i}.x"], 1e-5, f"stage{i}.x"); _close(c, g[f"stage{i}.c"], 1e-5, f"stage{i}.c")
    _close(logits, g["logits"], 1e-5, "logits")


def test_known_answers():
    """README.md:85-87,93-95: parameter counts 8.64 / 16.40 / 53.10 M (1000 classes), 8.33 / 16.04 / 52.61 M (51)."""
    assert O.count_params(O.VARIANTS["lemevit_tiny"], 1000) == 8_635_816
    assert O.count_params(O.VARIANTS["lemevit_small"], 1000) == 16_400_456
    assert O.count_params(O.VARIANTS["lemevit_base"], 1000) == 53_098_152
    assert O.count_params(O.VARIANTS["lemevit_base"], 51) == 52_611_315
    assert len(O.state_dict_spec(O.VARIANTS["lemevit_base"])) == 569
    # DCA scales quoted in SURVEY.md Appendix A for Base@224
    sx, sc = O.dca_scales(3136, 16, 96)
    assert abs(sx - 0.035149) < 1e-6 and abs(sc - 0.102062) < 1e-6


@pytest.mark.parametrize("name", ["train_tiny_96", "train_tiny_96_dp"])
def test_train_step(golden, name):
    meta, g = golden(name)
    cfg = O.VARIANTS[meta["variant"]]
    sd = fill_state_dict(O.state_dict_spec(cfg, meta["num_classes"]), meta["seed"])
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    img = det_tensor((meta["B"], 3, meta["res"], meta["res"]), name + ".img", 5)
    masks = None
    if "dp_masks" in g:
        # reference draw order: per block (skipping block 0 whose rate is 0 -> Identity), the
        # DropPath calls in source order (D/S: x-attn, x-mlp, c-attn, c-mlp; C: c-attn, c-mlp)
        masks, rows, r = {}, torch.from_numpy(g["dp_masks"]), 0
        first = True
        for i, (d, t) in enumerate(zip(cfg["depth"], cfg["attn_type"])):
            for j in range(d):
                if first:
                    first = False
                    continue
                n = 2 if t == "C" else 4
                masks[(i, j)] = [rows[r + q] for q in range(n)]
                r += n
        assert r == rows.shape[0]
    stats = {}
    logits = O.lemevit_forward(sd, cfg, img, train=True, dp_masks=masks, new_stats=stats)
    loss = torch.nn.functional.cross_entropy(logits, torch.tensor(meta["target"]))
    loss.backward()
    _close(logits.detach(), g["logits"], 1e-5, "logits")
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    gn = np.array([float(sd[k].grad.norm()) if sd[k].grad is not None else 0.0 for k in meta["param_names"]], dtype=np.float32)
    assert np.all(np.abs(gn - g["grad_norms"]) <= 3e-4 * np.maximum(1.0, np.abs(g["grad_norms"]))), np.abs(gn - g["grad_norms"]).max()
    for k in g:
        if k.startswith("grad.") and k != "grad_norms":
            _close(sd[k[5:]].grad, g[k], 1e-4, k)
        if k.startswith("stat."):
            _close(stats[k[5:]], g[k], 1e-5, k)


# ------------------------------------------------------------------------------------------------
# dense-prediction backbone: object_detection/mmdet/models/backbones/lemevit.py
def _dense_spec(cfg):
    spec = O.state_dict_spec(cfg, num_classes=0)
    for i, C in enumerate(cfg["embed_dim"][1:]):          # `extra_norms` (declared, never called) :759-762
        spec[f"extra_norms.{i}.weight"] = (C,); spec[f"extra_norms.{i}.bias"] = (C,)
    return spec


@pytest.mark.parametrize("name", ["dense_tiny_224", "dense_tiny_160x96"])
def test_dense_backbone_forward(golden, name):
    meta, g = golden(name)
    cfg = meta["cfg"]
    spec = _dense_spec(cfg)
    assert len(spec) == meta["nkeys"]
    sd = fill_state_dict(spec, meta["seed"])
    img = det_tensor((meta["B"], 3, meta["H"], meta["W"]), name + ".img", 6)
    with torch.no_grad():
        outs = O.lemevit_dense_forward(sd, cfg, img)
    assert [list(o.shape) for o in outs] == meta["shapes"]
    for i, o in enumerate(outs):
        _close(sample(o.flatten(2).transpose(1, 2), 8192), g[f"out{i}"], 1e-5, f"{name}.out{i}")


def test_dense_block_backward(golden):
    """The S block of the dense file: c comes back untouched and its gradient is the pass-through gradient only."""
    meta, g = golden("blockgrad_Sx")
    C, h, H, W, B = meta["C"], meta["h"], meta["H"], meta["W"], meta["B"]
    sd = {k: v.requires_grad_(True) for k, v in fill_state_dict(block_spec("S", C), meta["seed"]).items()}
    x = det_tensor((B, C, H, W), "blockgrad_Sx.x", 3).requires_grad_(True); c = det_tensor((B, 16, C), "blockgrad_Sx.c", 3).requires_grad_(True)
    gx = det_tensor((B, C, H, W), "blockgrad_Sx.gx", 3); gc = det_tensor((B, 16, C), "blockgrad_Sx.gc", 3)
    xt, _, _ = O.to_tokens(x)
    xo, co = O.leme_block(sd, "blk.", "Sx", xt, c, H, W, h)
    xo = O.to_nchw(xo, H, W)
    ((xo * gx).sum() + (co * gc).sum()).backward()
    _close(xo.detach(), g["x_out"], what="x_out"); _close(co.detach(), g["c_out"], what="c_out")
    assert np.array_equal(g["c_out"], c.detach().numpy()) and np.array_equal(g["dc"], gc.numpy())
    _close(x.grad, g["dx"], 1e-5, "dx"); _close(c.grad, g["dc"], 1e-5, "dc")
    for k, v in sd.items():
        gr = v.grad if v.grad is not None else torch.zeros_like(v)
        _close(gr, g["grad." + k[len("blk."):]], 2e-5, "grad " + k)
