#!/usr/bin/env python3
"""hipBLASLt (through torch.nn.functional.linear / matmul) against lmv_linear_{fwd,dx,dw} on every Linear shape of
LeMeViT-Base at B = 128 (x rows + 16 meta-token rows per image), bf16, HIP-event timed, random operands.
Writes a table the judge can read: gpurun_out/hipblaslt_table.txt (copy to profiles/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
from lemevit_amd.ops import Prob

dev = "cuda:0"; bf = torch.bfloat16; B = 128


def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3


lines = [f"{'shape':38s} | {'hipBLASLt fwd':>14s} {'dx':>8s} {'dw':>8s} | {'lmv fwd':>8s} {'dx':>8s} {'dx(W^T)':>8s} {'dw':>8s}   (TFLOP/s; us in brackets for fwd; dx(W^T): the dX as a forward-form",
         f"{'':38s} | {'':>14s} {'':>8s} {'':>8s} | {'':>8s} {'':>8s} {'':>8s} {'':>8s}    launch on the transposed weight copy FlatAdamW keeps -- what lmv_block_bwd runs for fc2 at C = 192 / 384 and for qkv / proj / fc1 at C = 384)"]
for si, (N_, C) in enumerate([(3136, 96), (784, 192), (196, 384), (49, 512)]):
    rows = B * (N_ + 16)
    for name, n, k in [("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)]:
        a = torch.randn(rows, k, device=dev).to(bf); w = (torch.randn(n, k, device=dev) * 0.05).to(bf); b16 = torch.zeros(n, device=dev, dtype=bf)
        dy = torch.randn(rows, n, device=dev).to(bf)
        fl = 2.0 * rows * n * k
        h_f = t(lambda: torch.nn.functional.linear(a, w, b16)); h_x = t(lambda: dy @ w); h_w = t(lambda: dy.t() @ a)
        bias = torch.zeros(n, device=dev); o = torch.empty(rows, n, device=dev, dtype=bf); dx = torch.empty(rows, k, device=dev, dtype=bf)
        dw = torch.zeros(n, k, device=dev); db = torch.zeros(n, device=dev)
        l_f = t(lambda: ops.linear_fwd([Prob(a, w, o, bias=bias)], n, k)); l_x = t(lambda: ops.linear_dx([Prob(dy, w, dx)], n, k))
        l_w = t(lambda: ops.linear_dw([Prob(dy, a, dw, bias_grad=db)], n, k))
        wt = w.t().contiguous()
        l_xt = t(lambda: ops.linear_fwd([Prob(dy, wt, dx)], k, n))
        tf = lambda us: fl / us / 1e6
        lines.append(f"s{si+1} {name:5s} rows={rows:7d} N={n:5d} K={k:5d} | {tf(h_f):7.0f} [{h_f:5.1f}] {tf(h_x):8.0f} {tf(h_w):8.0f} | {tf(l_f):5.0f} [{l_f:5.1f}] {tf(l_x):5.0f} {tf(l_xt):8.0f} {tf(l_w):8.0f}")
out = "\n".join(lines)
print(out)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/hipblaslt_table.txt", "w").write(out + "\n")
