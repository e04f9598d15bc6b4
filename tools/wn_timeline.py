#!/usr/bin/env python3
"""Timeline of one workgroup of the whole-width GEMM (csrc/wngemm.hip built with -DLMV_WN_TIMING; s_memtime ticks = 100 MHz constant clock
on gfx950 unless stated).  argv: rows K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rows, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (27136, 1536)
dbg = torch.zeros(256, device="cuda:0", dtype=torch.int64)
os.environ["LMV_WN_DBG_PTR"] = str(dbg.data_ptr())
from lemevit_amd import ops, _lib
from lemevit_amd.ops import Prob
_lib.config_set("gemm_wn", 2)
bf = torch.bfloat16; dev = "cuda:0"
a = torch.randn(rows, K, device=dev).to(bf); w = (torch.randn(384, K, device=dev) * 0.05).to(bf); o = torch.empty(rows, 384, device=dev, dtype=bf)
for _ in range(3):
    dbg.zero_()
    ops.linear_fwd([Prob(a, w, o)], 384, K, ops.ACT_NONE)
torch.cuda.synchronize()
d = dbg.cpu().view(2, 128)
for wgi, name in enumerate(("first workgroup", "middle workgroup")):
    t = [int(v) for v in d[wgi] if int(v) != 0]
    if len(t) < 6: print(name, "no stamps"); continue
    rel = [v - t[0] for v in t]
    nk = (len(t) - 6) // 2
    print(f"{name}: entry 0 | setup done {rel[1]} | first requests issued {rel[2]} | first k-step landed {rel[3]}")
    for k in range(nk):
        b = 4 + 2 * k
        print(f"   k-step {k:2d}: reads + mfma + requests +{t[b] - t[b - 1]:5d}   wait + barrier +{t[b + 1] - t[b]:5d}   (at {rel[b + 1]})")
    print(f"   epilogue issued +{t[-2] - t[-3]}   stores drained +{t[-1] - t[-2]}   total {rel[-1]} ticks")
