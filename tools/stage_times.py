"""Per-stage time of the inference forward's block schedule (the blocks of every stage on synthetic tokens of the stage's shape, bf16, no grad),
plus the stem / downsample layers: where the forward's milliseconds are.  usage: python tools/stage_times.py [model=lemevit_base] [B=128] [img=224]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lemevit_amd
from lemevit_amd import registry
import lemevit_amd.model as Mm

name = sys.argv[1] if len(sys.argv) > 1 else "lemevit_base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
IMG = int(sys.argv[3]) if len(sys.argv) > 3 else 224
dev = "cuda:0"
torch.manual_seed(0)
m = registry.create_model(name).to(dev).eval()
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x = torch.randn(B, 3, IMG, IMG, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    print(f"{name} B={B} whole forward: {timed(lambda: m(x)):.3f} ms")
    res = IMG // 4
    for i, stage in enumerate(m.stages):
        if i >= 2: res //= 2
        C = stage[0].norm1.weight.shape[0]
        xt = torch.randn(B, res * res, C, device=dev).bfloat16(); c = torch.randn(B, 16, C, device=dev).bfloat16()
        def run():
            a, b = xt, c
            whole = Mm._sstage_applies(stage, a, b, res, res)
            if whole is not None:
                return Mm._whole_stage_fwd(whole, a, b, Mm._sstage_packed(stage, whole), res, res)
            with Mm.image_ranges(a.device, B):
                for blk in stage:
                    a, b = blk.forward_tokens(a, b, res, res, masks=None)
            return a, b
        print(f"  stage {i}: {len(stage)} x {stage[0].kind} blocks, C = {C}, {res} x {res}: {timed(run):.3f} ms")
    # stem and stage transitions (models/lemevit.py:698-728) on synthetic feature maps of their input shapes
    res, cin = IMG, 3
    for i, ds in enumerate(m.downsample_layers):
        if isinstance(ds, torch.nn.Identity):
            continue
        if i == 0:
            inp = x
        else:
            Cp = m.stages[i - 1][0].norm1.weight.shape[0]  # (Base geometry below; other variants: the same resolutions)
            r = IMG // 4 if i <= 2 else (IMG // 8 if i == 3 else IMG // 16)
            inp = torch.randn(B, r, r, Cp, device=dev).bfloat16().permute(0, 3, 1, 2)
        def rund():
            y = m._run_downsample(ds, inp)
            return m._to_tokens(y, torch.bfloat16)
        print(f"  downsample {i} (input {tuple(inp.shape)}): {timed(rund):.3f} ms")
