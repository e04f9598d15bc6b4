#!/usr/bin/env python3
"""Tile-quantisation probe: time of the 384-wide stage-3 Linear launches against the row count (tiles of 128 x 128 per launch vs resident slots).
usage: python tools/quant_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
from lemevit_amd.ops import Prob
dev, bf = "cuda:0", torch.bfloat16

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for (N, K) in [(384, 1536), (384, 384), (1536, 384)]:
    print(f"--- N={N} K={K}: rows, tiles(128x128), fwd us, us per 1000 rows | dx us (out width K), per 1000 rows")
    for mt in (128, 160, 170, 171, 200, 212, 256, 300, 341, 342, 400, 512):
        rows = mt * 128
        a = torch.randn(rows, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.05).to(bf); bias = torch.zeros(N, device=dev)
        o = torch.empty(rows, N, device=dev, dtype=bf)
        dy = torch.randn(rows, N, device=dev).to(bf); dx = torch.empty(rows, K, device=dev, dtype=bf)
        tf = timeit(lambda: ops.linear_fwd([Prob(a, w, o, bias=bias)], N, K, ops.ACT_NONE))
        td = timeit(lambda: ops.linear_dx([Prob(dy, w, dx)], N, K))
        print(f"rows {rows:6d}  fwd tiles {mt * ((N + 127) // 128):5d}  {tf:7.1f} us  {tf / rows * 1e3:6.3f} | dx tiles {mt * ((K + 127) // 128):5d}  {td:7.1f} us  {td / rows * 1e3:6.3f}")
