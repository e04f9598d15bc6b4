#!/usr/bin/env python3
"""In-situ duration of every Linear launch shape of a Base train step (HIP events around each launch; blocks on the per-launch Python
schedule, weight gradients in line so that the events see one stream), for library switches given as KEY=VALUE ... on the command line
(lmv_config_set): the same process runs the step with the default configuration first, then with the switches applied.
usage: python tools/insitu_gemm.py gemm_wn=0 [gemm_rs=0 ...]"""
import os, sys, collections
os.environ["LMV_SIDE_STREAM"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lemevit_amd
from lemevit_amd import ops, _lib
import lemevit_amd.model as model, lemevit_amd.blocks as blocks
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = int(os.environ.get("B", "128"))
net = lemevit_amd.create_model("lemevit_base", num_classes=1000, drop_path_rate=0.1).to(dev).train()
opt = lemevit_amd.FlatAdamW(net, lr=1e-4, eps=1e-8, weight_decay=0.05)
x = torch.randn(B, 3, 224, 224, device=dev)
loss_fn = torch.nn.CrossEntropyLoss()
rec, on = collections.defaultdict(list), [False]
def wrap(name, fn):
    def timed(probs, N, K, *a, **kw):
        if not on[0]: return fn(probs, N, K, *a, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = fn(probs, N, K, *a, **kw); e.record()
        act = a[0] if a else kw.get("act", 0)
        key = (name, sum(p.rows for p in probs), N, K, int(act) if name != "dw" else 0, probs[0].res is not None, probs[0].out_pre is not None)
        rec[key].append((s, e))
        return r
    return timed
for name in ("linear_fwd", "linear_dx", "linear_dw"):
    t = wrap(name.split("_")[1], getattr(ops, name))
    for m in (ops, blocks.ops, model.ops): setattr(m, name, t)
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", torch.bfloat16):
        loss_fn(net(x), torch.randint(0, 1000, (B,), device=dev)).backward()
    opt.step()
def measure():
    rec.clear()
    model._NATIVE = False
    for _ in range(2): step()
    on[0] = True
    for _ in range(3): step()
    torch.cuda.synchronize(); on[0] = False
    return {k: (sum(s.elapsed_time(e) for s, e in v) / len(v) * 1e3, len(v) // 3) for k, v in rec.items()}
base = measure()
sw = [a.split("=") for a in sys.argv[1:]]
for k, v in sw: _lib.config_set(k, int(v))
alt = measure() if sw else None
tot = [0.0, 0.0]
print(f"{'kind':4s} {'rows':>7s} {'N':>5s} {'K':>5s} act res pre  n/step   default us" + ("   switched us" if alt else ""))
for k in sorted(base, key=lambda k: -base[k][0] * base[k][1]):
    us, n = base[k]
    line = f"{k[0]:4s} {k[1]:7d} {k[2]:5d} {k[3]:5d} {k[4]:3d} {int(k[5]):3d} {int(k[6]):3d}  {n:6d}   {us:10.1f}"
    tot[0] += us * n
    if alt and k in alt:
        line += f"   {alt[k][0]:10.1f}"; tot[1] += alt[k][0] * n
    print(line)
print(f"sum per step: default {tot[0] / 1e3:.3f} ms" + (f", switched {tot[1] / 1e3:.3f} ms ({' '.join(sys.argv[1:])})" if alt else ""))
