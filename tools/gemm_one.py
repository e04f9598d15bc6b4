import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemevit_amd import ops
from lemevit_amd.ops import Prob
dev = "cuda:0"; bf = torch.bfloat16
rows, n, k = 27136, 1536, 384
a = torch.randn(rows, k, device=dev).to(bf); w = (torch.randn(n, k, device=dev) * 0.05).to(bf); o = torch.empty(rows, n, device=dev, dtype=bf)
bias = torch.zeros(n, device=dev)
for _ in range(5): ops.linear_fwd([Prob(a, w, o, bias=bias)], n, k)
dy = torch.randn(rows, n, device=dev).to(bf); dx = torch.empty(rows, k, device=dev, dtype=bf)
for _ in range(5): ops.linear_dx([Prob(dy, w, dx)], n, k)
torch.cuda.synchronize()
