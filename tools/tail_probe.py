"""Does the time of an HBM-bound skinny GEMM (stage-1 fc1-dX: [rows, 384] x [384, 96]) follow the tile count or the round count?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
from lemevit_amd.ops import Prob
dev = "cuda:0"; bf = torch.bfloat16
n, k = 384, 96
w = (torch.randn(n, k, device=dev) * 0.05).to(bf)
for rows in [262144, 327680, 393216, 403456, 458752, 524288, 655360]:
    dy = torch.randn(rows, n, device=dev).to(bf); dx = torch.empty(rows, k, device=dev, dtype=bf)
    for _ in range(5): ops.linear_dx([Prob(dy, w, dx)], n, k)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(20): ops.linear_dx([Prob(dy, w, dx)], n, k)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    mb = rows * (n + k) * 2 / 1e6
    print(f"rows {rows:7d} tiles {rows // 128:5d} rounds {rows / 128 / 1024:5.2f}: {us:7.1f} us  {mb / us / 1e6 * 1e6 / 1e6:6.2f} TB/s  {us / (rows / 128) * 1e3:6.1f} ns/tile")
