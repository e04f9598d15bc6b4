import torch, time
dev='cuda:0'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e-3
for mb in (16, 83, 256, 1024):
    n = mb*1024*1024//2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); 
    tr = t(lambda: a.sum())
    print(f"{mb:5d} MB  fill {mb/1024/tf:7.2f} GB/ms={mb*1.048576e6/tf/1e12:5.2f} TB/s   copy(r+w) {2*mb*1.048576e6/tc/1e12:5.2f} TB/s   read(sum) {mb*1.048576e6/tr/1e12:5.2f} TB/s")
