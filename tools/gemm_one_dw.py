import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemevit_amd import ops
from lemevit_amd.ops import Prob
dev = "cuda:0"; bf = torch.bfloat16
rows, n, k = 27136, 1536, 384
dy = torch.randn(rows, n, device=dev).to(bf); x = torch.randn(rows, k, device=dev).to(bf)
dw = torch.zeros(n, k, device=dev); db = torch.zeros(n, device=dev)
for _ in range(5): ops.linear_dw([Prob(dy, x, dw, bias_grad=db)], n, k)
w = (torch.randn(n, k, device=dev) * 0.05).to(bf); dx = torch.empty(rows, k, device=dev, dtype=bf)
for _ in range(5): ops.linear_dx([Prob(dy, w, dx)], n, k)
torch.cuda.synchronize()
