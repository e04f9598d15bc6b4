#!/usr/bin/env python3
"""LayerNorm kernels alone on the Base B=128 block shapes (run under rocprofv3 --kernel-trace: tools/probe_prof.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
dev = "cuda:0"; bf = torch.bfloat16
B = 128
for (N, C) in [(3136, 96), (784, 192), (196, 384), (49, 512)]:
    x = torch.randn(B, N, C, device=dev).to(bf); c = torch.randn(B, 16, C, device=dev).to(bf)
    dyx = torch.randn_like(x); dyc = torch.randn_like(c)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    for _ in range(10):
        ys, st = ops.layernorm_fwd_multi([x, c], g, b, 1e-6, want_stats=True)
        dxs = ops.layernorm_bwd_multi([dyx, dyc], [x, c], st, g, dg, db, [dyx, dyc])
        z = x + dyx
    torch.cuda.synchronize()
