"""csrc/stem.hip (lmv_stem_fwd): the stem (conv 3x3/2 - BN - GELU - conv 3x3/2 - BN, models/lemevit.py:698-704) as ONE launch, against the pinned CPU
oracle's stem (float64, oracle/lemevit_oracle.py::lemevit_forward lines for stage 0) on the operands the kernel reads, and against the per-launch
im2col + GEMM schedule of the same library."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from detfill import det_tensor, fill_state_dict
from oracle import lemevit_oracle as O

DEV = "cuda:0"


def _stem_sd(Cm, Co, seed):
    spec = {"downsample_layers.0.0.weight": (Cm, 3, 3, 3), "downsample_layers.0.0.bias": (Cm,),
            "downsample_layers.0.1.weight": (Cm,), "downsample_layers.0.1.bias": (Cm,), "downsample_layers.0.1.running_mean": (Cm,), "downsample_layers.0.1.running_var": (Cm,),
            "downsample_layers.0.3.weight": (Co, Cm, 3, 3), "downsample_layers.0.3.bias": (Co,),
            "downsample_layers.0.4.weight": (Co,), "downsample_layers.0.4.bias": (Co,), "downsample_layers.0.4.running_mean": (Co,), "downsample_layers.0.4.running_var": (Co,)}
    return fill_state_dict(spec, seed)


def _oracle_stem(sd, img):
    sd = {k: v.double() for k, v in sd.items()}
    x = F.conv2d(img.double(), sd["downsample_layers.0.0.weight"], sd["downsample_layers.0.0.bias"], stride=2, padding=1)
    x = O.batch_norm(sd, "downsample_layers.0.1.", x, False)
    x = O.gelu_erf(x)
    x = F.conv2d(x, sd["downsample_layers.0.3.weight"], sd["downsample_layers.0.3.bias"], stride=2, padding=1)
    return O.batch_norm(sd, "downsample_layers.0.4.", x, False)


def _module(sd, Cm, Co):
    import torch.nn as nn
    seq = nn.Sequential(nn.Conv2d(3, Cm, 3, 2, 1), nn.BatchNorm2d(Cm), nn.GELU(), nn.Conv2d(Cm, Co, 3, 2, 1), nn.BatchNorm2d(Co))
    seq.load_state_dict({k[len("downsample_layers.0."):]: v for k, v in sd.items()}, strict=False)
    return seq.to(DEV).eval()


@pytest.mark.parametrize("Cm,Co", [(48, 96), (32, 64)])          # LeMeViT-Base / -Small, LeMeViT-Tiny
@pytest.mark.parametrize("B,H,W,layout", [(1, 32, 32, "nchw"), (3, 64, 96, "nchw"), (2, 224, 224, "nhwc"), (5, 224, 224, "bf16")])
def test_stem_vs_oracle(B, H, W, layout, Cm, Co):
    """Full maps against the float64 oracle (bf16 operands at both convolutions, the GELU polynomial: 1e-2 of max-abs, as the block kernels), every border case of
    both paddings (one-tile images, non-square maps), NCHW / channels-last / bf16 images; and the one-launch stem against the im2col + GEMM launches it replaces."""
    import lemevit_amd.model as Mm
    sd = _stem_sd(Cm, Co, 11)
    img = det_tensor((B, 3, H, W), "stem.img", 3)
    seq = _module(sd, Cm, Co)
    x = img.to(DEV)
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    if layout == "bf16":
        x = x.to(torch.bfloat16); img = x.float().cpu()
    mods = list(seq)
    with torch.no_grad():
        assert Mm._stem_applies(mods, x, torch.bfloat16)
        y = Mm._stem_fused(mods, x, torch.bfloat16)
        torch.cuda.synchronize()
        ref = _oracle_stem(sd, img)
        a, r = y.float().cpu().double().numpy(), ref.numpy()
        assert a.shape == r.shape and np.isfinite(a).all()
        err = float(np.abs(a - r).max() / np.abs(r).max())
        was, Mm._STEM = Mm._STEM, False
        try:
            with torch.autocast("cuda", torch.bfloat16):
                y2 = Mm.LeMeViT._run_downsample(_Holder(), seq, x)
        finally:
            Mm._STEM = was
        err2 = float((y.float() - y2.float()).abs().max() / y2.float().abs().max())
    print(f"stem {Cm}->{Co} B={B} {H}x{W} {layout}: vs oracle {err:.2e}, vs the per-launch schedule {err2:.2e}")
    assert err <= 1e-2 and err2 <= 2e-2, (err, err2)


class _Holder:
    training = False
