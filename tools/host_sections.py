#!/usr/bin/env python3
"""Host-side (launch) time of an eager train step by section: block forward calls, block backward calls, everything else."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lemevit_amd
import lemevit_amd.model as M
dev = torch.device("cuda:0")
model = lemevit_amd.create_model("lemevit_base", num_classes=1000, drop_path_rate=0.1).to(dev).train()
opt = lemevit_amd.FlatAdamW(model, lr=1e-4, weight_decay=0.05)
x = torch.randn(128, 3, 224, 224, device=dev); lf = torch.nn.CrossEntropyLoss()
acc = dict(fwd=0.0, bwd=0.0, nf=0, nb=0)
of, ob = M.block_forward, M.block_backward
def tf(*a, **k):
    t = time.perf_counter(); r = of(*a, **k); acc["fwd"] += time.perf_counter() - t; acc["nf"] += 1; return r
def tb(*a, **k):
    t = time.perf_counter(); r = ob(*a, **k); acc["bwd"] += time.perf_counter() - t; acc["nb"] += 1; return r
M.block_forward, M.block_backward = tf, tb
def step():
    t0 = time.perf_counter()
    opt.zero_grad()
    with torch.autocast("cuda", torch.bfloat16):
        out = model(x)
        t1 = time.perf_counter()
        loss = lf(out, torch.empty((128,), device=dev, dtype=torch.long).random_(1000))
        loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(5): step()
torch.cuda.synchronize()
for k in acc: acc[k] = 0
n = 10; tot = [0, 0, 0]
for _ in range(n):
    a, b, c = step(); tot[0] += a; tot[1] += b; tot[2] += c
torch.cuda.synchronize()
print(f"per step (ms): forward call {1e3*tot[0]/n:.2f} (of which block_forward {1e3*acc['fwd']/n:.2f} over {acc['nf']//n} blocks), "
      f"loss+backward call {1e3*tot[1]/n:.2f} (block_backward {1e3*acc['bwd']/n:.2f}), optimizer {1e3*tot[2]/n:.2f}")
