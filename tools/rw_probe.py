#!/usr/bin/env python3
"""The C = 96 one-kernel MLP on the stage-1 shape (403 456 rows): resident-weight form (csrc/rwmlp.hip) against the tile-streaming form
(LMV_MLP_RW96 through lmv_config_set), HIP-event timed.  usage: rw_probe.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops, _lib
dev = "cuda:0"; bf = torch.bfloat16
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 3136
x = torch.randn(128, rows // 128, 96, device=dev).to(bf)
w1 = torch.randn(384, 96, device=dev) * 0.1; b1 = torch.randn(384, device=dev) * 0.1
g = torch.rand(96, device=dev) + 0.5; be = torch.randn(96, device=dev) * 0.1
w2 = (torch.randn(96, 384, device=dev) * 0.05).to(bf); b2 = torch.randn(96, device=dev) * 0.1
F = ops.ln_fold(w1, b1, g, be, bf)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
_lib.config_set("mlp_rw96", 1)
print(f"resident-weight form {t(lambda: ops.mlp_fused_fwd([x], F, w2, b2, 1e-6)):7.1f} us", flush=True)
_lib.config_set("mlp_rw96", 0)
print(f"tile-streaming form {t(lambda: ops.mlp_fused_fwd([x], F, w2, b2, 1e-6)):7.1f} us")
