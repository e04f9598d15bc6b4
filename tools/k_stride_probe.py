import os, sys
sys.path.insert(0, os.getcwd())
import torch
from lemevit_amd import ops, _lib
from lemevit_amd.ops import Prob
dev, bf = "cuda:0", torch.bfloat16
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
rows = 27136
for K in (1280, 1344, 1408, 1472, 1536, 1600, 1664, 1728, 2048, 2112):
    a = torch.randn(rows, K, device=dev).to(bf); w = (torch.randn(384, K, device=dev) * 0.05).to(bf); o = torch.empty(rows, 384, device=dev, dtype=bf)
    r = []
    for mode in (0, 2):
        _lib.config_set("gemm_wn", mode)
        t = timeit(lambda: ops.linear_fwd([Prob(a, w, o)], 384, K, ops.ACT_NONE))
        r.append(t)
    print(f"K {K:5d} (row stride {2*K:5d} B = {2*K/256:6.2f} x 256): tile {r[0]:6.1f} us {2.0*rows*384*K/r[0]/1e6:6.0f} TF | wn {r[1]:6.1f} us {2.0*rows*384*K/r[1]/1e6:6.0f} TF   ns per k64-step: wn {r[1]*1e3/(K/64):6.0f}")
