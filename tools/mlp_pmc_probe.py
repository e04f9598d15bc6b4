#!/usr/bin/env python3
"""One fused MLP launch shape, repeated (for rocprofv3 --pmc passes, tools/pmc_mlp.sh).  argv: B N C  (default 128 3136 96)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
dev = "cuda:0"; bf = torch.bfloat16
B, N, C = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (128, 3136, 96)
Hd = 4 * C
x = torch.randn(B, N, C, device=dev).to(bf); c = torch.randn(B, 16, C, device=dev).to(bf)
g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
w1 = torch.randn(Hd, C, device=dev) * 0.05; b1 = torch.zeros(Hd, device=dev)
w2 = (torch.randn(C, Hd, device=dev) * 0.05).to(bf); b2 = torch.zeros(C, device=dev)
F = ops.ln_fold(w1, b1, g, be, bf)
for _ in range(10):
    ops.mlp_fused_fwd([x, c], F, w2, b2, 1e-6)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", B * (N + 16) * C * 4 + 4 * C * Hd, " flops:", 4 * B * (N + 16) * C * Hd)
