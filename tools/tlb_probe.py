#!/usr/bin/env python3
"""Is the cold-operand slowdown of the Linear launches a TLB effect or an HBM effect?  fc2-forward shape, operands rotating over 6 sets
(750 MB: out of the MALL), (a) untouched, (b) with one dword per 4 KB page of the NEXT set read by a tiny kernel first (translations warm,
data still cold), (c) with the whole next set read first (data warm in the MALL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops, _lib
from lemevit_amd.ops import Prob
dev, bf = "cuda:0", torch.bfloat16
rows, K, nbuf = 27136, 1536, 6
sets = [(torch.randn(rows, K, device=dev).to(bf), torch.randn(rows, 384, device=dev).to(bf), torch.empty(rows, 384, device=dev, dtype=bf)) for _ in range(nbuf)]
w = (torch.randn(384, K, device=dev) * 0.05).to(bf); bias = torch.zeros(384, device=dev)
sink = torch.zeros(1, device=dev)
def touch(i, mode):
    if mode == 0: return
    for t in sets[i % nbuf]:
        f = t.view(-1)
        sink.add_(f[::2048].float().sum() if mode == 1 else f.float().sum())
def run(mode, wn):
    _lib.config_set("gemm_wn", wn)
    tot, n = 0.0, 4 * nbuf
    for i in range(2 * nbuf + n):
        a, r, o = sets[i % nbuf]
        touch(i, mode)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.linear_fwd([Prob(a, w, o, bias=bias, res=r)], 384, K, ops.ACT_NONE); e.record()
        torch.cuda.synchronize()
        if i >= 2 * nbuf: tot += s.elapsed_time(e)
    return tot / n * 1e3
for wn in (0, 2):
    print(f"gemm_wn={wn}: cold {run(0, wn):6.1f} us | translations warmed {run(1, wn):6.1f} us | data warmed {run(2, wn):6.1f} us")
_lib.config_set("gemm_wn", 1)
