#!/usr/bin/env python3
"""Timeline of one workgroup of a Linear GEMM launch (s_memtime stamps; library built with make CXXEXTRA=-DLMV_GEMM_TIMING).
argv: rows N K [fwd|dx|dw]   (default: stage-3 fc1 forward of Base at B = 128)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rows, N, K = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (27136, 1536, 384)
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
plain = os.environ.get("PLAIN", "0") == "1"          # forward without bias / GELU: is the epilogue's time VALU or the store path?
dbg = torch.zeros(512, device="cuda:0", dtype=torch.int64)
os.environ["LMV_GEMM_DBG_PTR"] = str(dbg.data_ptr())
from lemevit_amd import ops
from lemevit_amd.ops import Prob
bf = torch.bfloat16; dev = "cuda:0"
a = torch.randn(rows, K if mode == "fwd" else N, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
xx = torch.randn(rows, K, device=dev).to(bf); dwo = torch.zeros(N, K, device=dev); dbo = torch.zeros(N, device=dev)
bias = torch.zeros(N, device=dev)
out = torch.empty(rows, N if mode == "fwd" else K, device=dev, dtype=bf)
for _ in range(3):
    dbg.zero_()
    if mode == "fwd": ops.linear_fwd([Prob(a, w, out, bias=None if plain else bias)], N, K, ops.ACT_NONE if plain else ops.ACT_GELU)
    elif mode == "dx": ops.linear_dx([Prob(a, w, out)], N, K)
    else: ops.linear_dw([Prob(a, xx, dwo, bias_grad=dbo)], N, K)
torch.cuda.synchronize()
d = dbg.cpu().view(2, 256)
for wgi, name in enumerate(("first workgroup", "middle workgroup")):
    t = [int(v) for v in d[wgi] if int(v) != 0]
    if len(t) < 6: print(name, "no stamps"); continue
    t0 = t[0]; rel = [v - t0 for v in t]
    nk = K // 64 if (K % 64 == 0 and mode == "fwd") else (K if mode == "fwd" else N) // 32
    if mode == "dw": nk = (len(t) - 3) // 3
    elif 3 + 3 * nk + 3 > len(t): nk = (len(t) - 6) // 3
    print(f"{name}: entry 0 | requests issued {rel[1]} | first k-tile landed {rel[2]}")
    for k in range(nk):
        b = 3 + 3 * k
        print(f"   k-tile {k}: request +{t[b] - t[b - 1]:5d}  reads+mfma +{t[b + 1] - t[b]:5d}  wait+barrier +{t[b + 2] - t[b + 1]:5d}   (at {rel[b + 2]})")
    if mode == "dw":
        print(f"   k-loop end {rel[2 + 3 * nk]} ({nk} k-tiles, {(rel[2 + 3 * nk] - rel[2]) / max(nk, 1):.0f} cycles per k-tile)")
        continue
    e = 3 + 3 * nk
    tail = t[e + 1:]
    print(f"   k-loop end {rel[e]} | epilogue operands requested +{t[e + 1] - t[e]} | then (LDS transpose done, then each 16-byte store issued): " + " ".join(f"+{b - a}" for a, b in zip(tail, tail[1:])) + f" | total {rel[-1]} cycles")
