import torch, time
dev="cuda:0"; bf=torch.bfloat16
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
for rows,n,k in [(403456,288,96),(403456,384,96),(403456,96,384),(102400,768,192),(27136,1152,384),(27136,1536,384),(27136,384,1536),(8320,2048,512),(8192,8192,8192)]:
    a=torch.randn(rows,k,device=dev).to(bf); w=(torch.randn(n,k,device=dev)*0.05).to(bf); b=torch.zeros(n,device=dev,dtype=bf)
    us=t(lambda: torch.nn.functional.linear(a,w,b))
    us2=t(lambda: a.t() @ torch.randn(1,1,device=dev).to(bf).expand(rows, 1)) if False else 0
    dy=torch.randn(rows,n,device=dev).to(bf)
    us_dw=t(lambda: dy.t() @ a)
    us_dx=t(lambda: dy @ w)
    fl=2.0*rows*n*k
    print(f"rows={rows:7d} N={n:5d} K={k:5d}: torch fwd {us:8.1f} us {fl/us/1e6:7.1f} TF | dx {us_dx:8.1f} us {fl/us_dx/1e6:7.1f} TF | dw {us_dw:8.1f} us {fl/us_dw/1e6:7.1f} TF")
