#!/usr/bin/env python3
"""dwconv kernels alone (run under rocprofv3 --kernel-trace for true kernel durations); a torch copy of the same map for scale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
dev = "cuda:0"; bf = torch.bfloat16
B = 128
for (H, C) in [(56, 96), (28, 192), (14, 384), (7, 512)]:
    x = torch.randn(B, H * H, C, device=dev).to(bf); dy = torch.randn_like(x)
    w = torch.randn(C, 1, 3, 3, device=dev) * 0.1; b = torch.zeros(C, device=dev)
    dw = torch.zeros_like(w); db = torch.zeros_like(b)
    for _ in range(10):
        y = ops.dwconv_residual_fwd(x, w, b, H, H)
        dx = ops.dwconv_residual_bwd_data(dy, w, H, H)
        ops.dwconv_bwd_weight(dy, x, dw, db, H, H)
        z = x.clone()
        z2 = x + dy
    torch.cuda.synchronize()
