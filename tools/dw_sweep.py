#!/usr/bin/env python3
"""dW GEMM: sweep the k-split slot count and k-tile depth per shape (one process; the library re-reads the env per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops, _lib
from lemevit_amd.ops import Prob
dev = "cuda:0"; bf = torch.bfloat16
B = 128
SHAPES = []
for si, (N, C) in enumerate([(3136, 96), (784, 192), (196, 384), (49, 512)]):
    for name, n, k in [("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)]:
        SHAPES.append((f"s{si+1} {name}", B * N, B * 16, n, k))

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

slots = [256, 384, 512, 640, 768, 1024, 1536, 2048, 3072]
print(f"{'shape':10s} bk " + " ".join(f"{s:>6d}" for s in slots))
for name, rx, rc, n, k in SHAPES:
    ax = torch.randn(rx, k, device=dev).to(bf); ac = torch.randn(rc, k, device=dev).to(bf)
    dyx = torch.randn(rx, n, device=dev).to(bf); dyc = torch.randn(rc, n, device=dev).to(bf)
    dw = torch.zeros(n, k, device=dev); db = torch.zeros(n, device=dev)
    for bk in (64, 32):
        _lib.config_set("dw_bk", bk)
        row = []
        for s in slots:
            _lib.config_set("dw_target_blocks", s)
            row.append(timeit(lambda: ops.linear_dw([Prob(dyx, ax, dw, bias_grad=db), Prob(dyc, ac, dw, bias_grad=db)], n, k)))
        print(f"{name:10s} {bk:2d} " + " ".join(f"{t:6.1f}" for t in row), flush=True)
