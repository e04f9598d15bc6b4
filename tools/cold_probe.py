#!/usr/bin/env python3
"""Whole-width vs tile kernel on the fc2 / proj forward shapes with the token operand NOT resident in the 256 MB MALL: the launch rotates
over NBUF (A, residual, out) sets (NBUF x 125 MB), as in the model, where fc1 has just written 166 MB before fc2 reads 83 MB of it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops, _lib
from lemevit_amd.ops import Prob
dev, bf = "cuda:0", torch.bfloat16
rows = 27136
def run(K, nbuf, mode, res):
    _lib.config_set("gemm_wn", mode)
    sets = [(torch.randn(rows, K, device=dev).to(bf), torch.randn(rows, 384, device=dev).to(bf), torch.empty(rows, 384, device=dev, dtype=bf)) for _ in range(nbuf)]
    w = (torch.randn(384, K, device=dev) * 0.05).to(bf); bias = torch.zeros(384, device=dev)
    def go(i):
        a, r, o = sets[i % nbuf]
        ops.linear_fwd([Prob(a, w, o, bias=bias, res=r if res else None)], 384, K, ops.ACT_NONE)
    for i in range(2 * nbuf): go(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 8 * nbuf
    s.record()
    for i in range(n): go(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for K in (1536, 384):
    for nbuf in (1, 6):
        t0, t2 = run(K, nbuf, 0, True), run(K, nbuf, 2, True)
        print(f"K {K:4d}  {nbuf} operand set(s): tile {t0:6.1f} us   whole-width {t2:6.1f} us")
_lib.config_set("gemm_wn", 1)
