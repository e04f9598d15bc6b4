"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/lemevit_hip.h
declares (no compute calls without a GPU), the host-side interface mirrors the reference's, and the product path
refuses to run without the HIP kernels."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lemevit_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lmv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from lemevit_amd import _lib
    names = _declared()
    assert len(names) >= 19
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/lemevit_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert _lib.lib.lmv_abi_version() == _lib.ABI_VERSION


def test_error_reporting_without_gpu():
    """Argument validation happens before any launch: bad shapes return LMV_ERR_SHAPE with a message, nothing throws."""
    from lemevit_amd._lib import lib, LinearProblem
    p = (LinearProblem * 1)()
    assert lib.lmv_linear_fwd(p, 3, 64, 64, 0, 1, None) == -1 and b"nproblems" in lib.lmv_last_error()
    assert lib.lmv_linear_fwd(p, 1, 60, 64, 0, 1, None) == -1 and b"multiples of 8" in lib.lmv_last_error()
    assert lib.lmv_linear_fwd(p, 1, 64, 64, 0, 7, None) == -2
    assert lib.lmv_layernorm_fwd(None, 1, None, None, 12, 1e-6, 0, None) == -1
    assert lib.lmv_attn_workspace_bytes(128, 3, 16, 3136, 0) == 128 * 3 * 25 * 16 * 34 * 4      # 25 key ranges of 128
    assert lib.lmv_sa_core_fwd(None, None, None, 1, 4, 48, None, 0, 0, None) == -1 and b"head dim" in lib.lmv_last_error()


def test_registry_and_state_dict_layout():
    import lemevit_amd as L
    from oracle import lemevit_oracle as O
    assert {"lemevit_tiny", "lemevit_small", "lemevit_base", "lemevit_small_v2", "lemevit_tiny_v2", "vit_tiny"} <= set(L.list_models())
    # timm-style call as benchmark.py:409-419 makes it (None-valued kwargs are dropped)
    m = L.create_model("lemevit_tiny", pretrained=False, num_classes=51, in_chans=3, global_pool=None, scriptable=False,
                       drop_rate=0.0, drop_path_rate=0.1, drop_block_rate=None)
    spec = O.state_dict_spec(O.VARIANTS["lemevit_tiny"], 51)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in sd)
    assert sum(p.numel() for p in m.parameters()) == 8_331_187        # README.md:93
    assert m.num_classes == 51 and m.default_cfg["input_size"] == (3, 224, 224) and m.get_classifier() is m.head
    assert m.no_weight_decay() == {"pos_embed", "cls_token"}
    # DropPath rates: linspace(0, rate, sum(depth)), block 0 is identity (models/lemevit.py:750,531)
    rates = [b.drop_prob for st in m.stages for b in st]
    assert rates[0] == 0.0 and abs(rates[-1] - 0.1) < 1e-7 and rates == sorted(rates)
    with pytest.raises(RuntimeError):
        L.create_model("resnet50")


def test_checkpoint_formats(tmp_path):
    import lemevit_amd as L
    a = L.create_model("lemevit_tiny", num_classes=10)
    for wrap, prefix in [("model", ""), ("state_dict", "module."), (None, "backbone.")]:
        sd = {prefix + k: v for k, v in a.state_dict().items()}
        path = str(tmp_path / f"ck_{wrap}.pth")
        torch.save({wrap: sd} if wrap else sd, path)
        b = L.create_model("lemevit_tiny", num_classes=10, checkpoint_path=path)
        assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
    c = L.lemevit_tiny(pretrained=str(tmp_path / "ck_model.pth"), num_classes=10)     # reference: pretrained=<path>
    assert torch.equal(c.head.weight, a.head.weight)


def test_no_cpu_fallback():
    import lemevit_amd as L
    m = L.create_model("lemevit_tiny", num_classes=10).eval()
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.randn(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="GPU"):
        L.ops.layernorm_fwd(torch.randn(4, 64), torch.ones(64), torch.zeros(64), 1e-6)


def test_dense_backbone_checkpoint_remap(tmp_path):
    """init_weights() of the dense-prediction backbone accepts the reference's checkpoint layouts: {'state_dict': ...} with
    'backbone.' / 'module.' prefixes (object_detection/mmdet/models/backbones/lemevit.py:844-877); the classifier's head keys are
    reported as unexpected, not loaded."""
    import torch
    import lemevit_amd
    cfg = dict(depth=[1, 1, 1, 1, 1], embed_dim=[64, 64, 128, 192, 320], head_dim=32, attn_type=["C", "D", "D", "S", "S"], queries_len=16)
    src = lemevit_amd.LeMeViT(num_classes=10, **cfg)
    for p in src.parameters():
        torch.nn.init.normal_(p, std=0.02)
    path = tmp_path / "cls.pth.tar"
    torch.save({"state_dict": {"module.backbone." + k: v for k, v in src.state_dict().items()}}, path)
    bb = lemevit_amd.LeMeViTBackbone(pretrained=str(path), **cfg)
    got, want = bb.state_dict(), src.state_dict()
    assert all(torch.equal(got[k], want[k]) for k in want if not k.startswith("head."))
    res = bb.init_weights(str(path))
    assert sorted(res.unexpected_keys) == ["head.bias", "head.weight"] and all(k.startswith("extra_norms.") for k in res.missing_keys)
    assert not hasattr(bb, "head") and all(blk.kind == "Sx" for st in bb.stages[3:] for blk in st)


def test_schedule_helpers_without_gpu():
    """Host logic of the concurrent schedules that does not need a device: split_forward with one part is the plain call (and fills `outs`),
    image_ranges is a no-op off the GPU / under no_grad / for small batches, the range entry point is bound with the header's signature."""
    from lemevit_amd import _lib
    from lemevit_amd.graph import split_forward
    import lemevit_amd.model as M
    calls = []

    def f(t):
        calls.append(t.shape[0])
        return t * 2

    x = torch.arange(12.0).view(6, 2)
    assert torch.equal(split_forward(f, x, 1), x * 2) and calls == [6]
    outs = []
    split_forward(f, x, 0, outs)                      # parts < 1 is clamped to 1
    assert len(outs) == 1 and torch.equal(outs[0], x * 2)
    cpu = torch.device("cpu")
    with M.image_ranges(cpu, 128) as r:
        assert not r.on and not M._range_state
    with torch.no_grad():
        assert not M.image_ranges(torch.device("cuda", 0), 128).on
    assert not M.image_ranges(torch.device("cuda", 0), M.TRAIN_PARTS_MIN_BATCH - 1).on
    assert not M.image_ranges(torch.device("cuda", 0), 128, parts=1).on
    fn = _lib.lib.lmv_block_fwd_range
    assert fn.restype is ctypes.c_int and len(fn.argtypes) == 11
