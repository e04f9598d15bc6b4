#!/usr/bin/env python3
"""One forward Linear GEMM shape, repeated (for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes).
Default: stage-3 fc1 of LeMeViT-Base at B=128 (rows 25088 + 2048, N=1536, K=384); argv: rows_x rows_c N K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
from lemevit_amd.ops import Prob
dev = "cuda:0"; bf = torch.bfloat16
rx, rc, n, k = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (25088, 2048, 1536, 384)
mode = sys.argv[5] if len(sys.argv) > 5 else "fwd"          # fwd | dx | dw
ax = torch.randn(rx, k, device=dev).to(bf); ac = torch.randn(rc, k, device=dev).to(bf)
w = (torch.randn(n, k, device=dev) * 0.05).to(bf); bias = torch.zeros(n, device=dev)
ox = torch.empty(rx, n, device=dev, dtype=bf); oc = torch.empty(rc, n, device=dev, dtype=bf)
dxx = torch.empty_like(ax); dxc = torch.empty_like(ac)
dw = torch.zeros(n, k, device=dev); db = torch.zeros(n, device=dev)
for _ in range(20):
    if mode == "fwd":
        ops.linear_fwd([Prob(ax, w, ox, bias=bias), Prob(ac, w, oc, bias=bias)], n, k)
    elif mode == "dx":
        ops.linear_dx([Prob(ox, w, dxx), Prob(oc, w, dxc)], n, k)
    else:
        ops.linear_dw([Prob(ox, ax, dw, bias_grad=db), Prob(oc, ac, dw, bias_grad=db)], n, k)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", (rx + rc) * (n + k) * 2 + n * k * 2, " flops:", 2 * (rx + rc) * n * k)
