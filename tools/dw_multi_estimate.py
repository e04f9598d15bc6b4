#!/usr/bin/env python3
"""Would ONE weight-gradient launch per block pay?  The four dW problems of a stage-3 S block (fc2, fc1, proj, qkv; x rows + c rows each) as today -- four launches in a row on one
stream, each split to ~768 workgroups, each followed by its slab reduction -- against the same four launched CONCURRENTLY on four streams with a quarter of the split target each
(what a fused launch would look like to the chip: 108 tiles x ~7 splits in flight together, a quarter of the slab bytes), reductions behind them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops, _lib
from lemevit_amd.ops import Prob
dev = "cuda:0"; bf = torch.bfloat16
B, N, M, C = 128, 196, 16, 384
rx, rc = B * N, B * M
shapes = [("fc2", C, 4 * C), ("fc1", 4 * C, C), ("proj", C, C), ("qkv", 3 * C, C)]
ops_ = []
for name, n, k in shapes:
    ax = torch.randn(rx, k, device=dev).to(bf); ac = torch.randn(rc, k, device=dev).to(bf)
    dyx = torch.randn(rx, n, device=dev).to(bf); dyc = torch.randn(rc, n, device=dev).to(bf)
    dw = torch.zeros(n, k, device=dev); db = torch.zeros(n, device=dev)
    ops_.append((name, n, k, [Prob(dyx, ax, dw, bias_grad=db), Prob(dyc, ac, dw, bias_grad=db)]))
streams = [torch.cuda.Stream() for _ in ops_]

def serial():
    for name, n, k, p in ops_:
        ops.linear_dw(p, n, k)

def concurrent():
    cur = torch.cuda.current_stream()
    for s, (name, n, k, p) in zip(streams, ops_):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            ops.linear_dw(p, n, k)
    for s in streams:
        cur.wait_stream(s)

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

_lib.config_set("dw_target_blocks", 0)
print(f"today: four launches in a row, ~768 workgroups each (+ reductions): {timeit(serial):.1f} us")
for tgt in (768, 384, 256, 192, 128):
    _lib.config_set("dw_target_blocks", tgt)
    print(f"four launches concurrently, split target {tgt:4d} each: {timeit(concurrent):.1f} us;  in a row: {timeit(serial):.1f} us")
_lib.config_set("dw_target_blocks", 0)
