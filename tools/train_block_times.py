#!/usr/bin/env python3
"""Per-block wall time of the Base train step on the launch stream (forward and backward separately): HIP events recorded before every block's forward call and when the
gradient of every block's output arrives (tensor hooks).  Eager step as bench.py runs it (B = 128, bf16 autocast, drop_path 0.1, FlatAdamW); image ranges off so that a
block's launches are one stream's.  Usage: train_block_times.py [model] [batch]"""
import os, sys
os.environ.setdefault("LMV_TRAIN_PARTS", "1")
import torch
sys.path.insert(0, ".")
import lemevit_amd
import lemevit_amd.model as M

name = sys.argv[1] if len(sys.argv) > 1 else "lemevit_base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = lemevit_amd.create_model(name, num_classes=1000, drop_path_rate=0.1).to(dev).train()
opt = lemevit_amd.FlatAdamW(model, lr=1e-4, eps=1e-8, weight_decay=0.05)
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (B,), device=dev)
lossf = torch.nn.CrossEntropyLoss()
rec = None
blocks = [(si, bi, blk) for si, st in enumerate(model.stages) for bi, blk in enumerate(st)]
ids = {id(b): (si, bi) for si, bi, b in blocks}
orig = M.LeMeBlock.forward_tokens


def patched(self, xt, c, H, W, masks=None, prev=None):
    if rec is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(); rec["f"].append((ids[id(self)], e))
    xo, co = orig(self, xt, c, H, W, masks=masks, prev=prev)
    if rec is not None and co.requires_grad:
        key = ids[id(self)]

        def hook(g, key=key):
            e = torch.cuda.Event(enable_timing=True); e.record(); rec["b"].append((key, e)); return g
        co.register_hook(hook)
    return xo, co


M.LeMeBlock.forward_tokens = patched


def step():
    with torch.autocast("cuda", torch.bfloat16):
        loss = lossf(model(x), y)
    if rec is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(); rec["f"].append((("loss", 0), e))
    loss.backward()
    if rec is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(); rec["b"].append((("input", 0), e))
    opt.step(); opt.zero_grad(set_to_none=False)
    if rec is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(); rec["b"].append((("opt", 0), e))


for _ in range(5):
    step()
torch.cuda.synchronize()
acc = {}
NREP = 5
for _ in range(NREP):
    rec = {"f": [], "b": []}
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    step()
    torch.cuda.synchronize()
    seq = [(("start", 0), e0)] + rec["f"] + rec["b"]
    for (ka, ea), (kb, eb) in zip(seq[:-1], seq[1:]):
        # the interval that STARTS at event a: forward list = the block's forward; backward list: the hook of block k fires when its output gradient is ready = the start of ITS backward
        phase = "fwd" if (ka, ea) in rec["f"] or ka[0] == "start" else "bwd"
        acc.setdefault((phase, ka), 0.0)
        acc[(phase, ka)] += ea.elapsed_time(eb) / NREP
    total = e0.elapsed_time(seq[-1][1])
rec = None
print(f"{name} B={B}: step {total:.2f} ms (last repetition, with events, one range of images)")
print("phase,stage,block,ms")
per_stage = {}
for (phase, key), ms in acc.items():
    print(f"{phase},{key[0]},{key[1]},{ms:.3f}")
    per_stage.setdefault((phase, key[0]), 0.0)
    per_stage[(phase, key[0])] += ms
print("per stage (ms): " + ", ".join(f"{p} {s}: {v:.2f}" for (p, s), v in per_stage.items()))
