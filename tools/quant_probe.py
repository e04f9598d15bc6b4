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

from lemevit_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] == "wn":          # whole-width kernel (csrc/wngemm.hip) against the tile kernels on the 384-wide launches
    for K in (384, 1152, 1536):
        for rows in (27136, 8320, 102400):
            a = torch.randn(rows, K, device=dev).to(bf); w = (torch.randn(384, K, device=dev) * 0.05).to(bf); bias = torch.zeros(384, device=dev)
            o = torch.empty(rows, 384, device=dev, dtype=bf); res = torch.randn(rows, 384, device=dev).to(bf)
            t = {}
            for mode in (0, 2):
                _lib.config_set("gemm_wn", mode)
                t[mode] = (timeit(lambda: ops.linear_fwd([Prob(a, w, o, bias=bias)], 384, K, ops.ACT_NONE)),
                           timeit(lambda: ops.linear_fwd([Prob(a, w, o, bias=bias, res=res)], 384, K, ops.ACT_NONE)))
            _lib.config_set("gemm_wn", 1)
            fl = 2.0 * rows * 384 * K
            print(f"rows {rows:6d} K {K:4d}: tile {t[0][0]:6.1f} us ({fl / t[0][0] / 1e6:6.0f} TF)  wn {t[2][0]:6.1f} us ({fl / t[2][0] / 1e6:6.0f} TF) | +res: tile {t[0][1]:6.1f}  wn {t[2][1]:6.1f}")
    sys.exit(0)
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
