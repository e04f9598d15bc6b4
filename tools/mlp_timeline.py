#!/usr/bin/env python3
"""Per-step timeline of ONE workgroup of the fused MLP kernel (s_memtime stamps; needs a library built with -DLMV_MLP_TIMING:
   make -C lemevit_amd/csrc clean && make -C lemevit_amd/csrc CXXEXTRA=-DLMV_MLP_TIMING).  argv: C [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
C = int(sys.argv[1]) if len(sys.argv) > 1 else 384
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 27136
dbg = torch.zeros(8192, device="cuda:0", dtype=torch.int64)
os.environ["LMV_MLP_DBG_PTR"] = str(dbg.data_ptr())
from lemevit_amd import ops
bf = torch.bfloat16; dev = "cuda:0"; Hd = 4 * C
x = torch.randn(1, rows, C, device=dev).to(bf)
g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
F = ops.ln_fold(torch.randn(Hd, C, device=dev) * 0.05, torch.zeros(Hd, device=dev), g, be, bf)
w2 = (torch.randn(C, Hd, device=dev) * 0.05).to(bf); b2 = torch.zeros(C, device=dev)
for _ in range(3):
    ops.mlp_fused_fwd([x], F, w2, b2, 1e-6)
torch.cuda.synchronize()
d = dbg.cpu().view(2, 4096)
for w in range(2):
    t = d[w][: 800 * 5].view(800, 5)
    n = int((t[:, 0] != 0).sum())
    t = t[:n].double()
    print(f"wave {'0' if w == 0 else 'last'}: {n} steps; per-step cycles (s_memtime ticks at 100 MHz x ... shown raw): ")
    iss, mma, vm, bar = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    tot = t[1:, 0] - t[:-1, 0]
    print("  issue %.1f  reads+mfma %.1f  wait_vm %.1f  barrier %.1f  step-to-step %.1f (median ticks)" % (iss.median(), mma.median(), vm.median(), bar.median(), tot.median()))
    k = min(n, 14)
    for i in range(k):
        print("   step %2d: issue %5.0f mma %5.0f vm %5.0f bar %5.0f | next %5.0f" % (i, iss[i], mma[i], vm[i], bar[i], (tot[i] if i < n - 1 else 0)))
    print("  whole loop: %.0f ticks for %d steps" % (t[-1, 4] - t[0, 0], n))
