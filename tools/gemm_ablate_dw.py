import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemevit_amd import ops
from lemevit_amd.ops import Prob
dev = "cuda:0"; bf = torch.bfloat16
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters*1e3
for rows, n, k in [(27136, 1536, 384), (27136, 384, 384), (403456, 384, 96), (102400, 768, 192)]:
    dy = torch.randn(rows, n, device=dev).to(bf); x = torch.randn(rows, k, device=dev).to(bf)
    dw = torch.zeros(n, k, device=dev); db = torch.zeros(n, device=dev)
    t = timeit(lambda: ops.linear_dw([Prob(dy, x, dw, bias_grad=db)], n, k))
    t2 = timeit(lambda: ops.linear_dw([Prob(dy, x, dw)], n, k))
    print(f"rows={rows} N={n} K={k}: {t:.1f} us {2.0*rows*n*k/t/1e6:.0f} TF | no-bias-grad {t2:.1f} us")
