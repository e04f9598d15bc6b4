import torch,time
torch.cuda.synchronize()
for n in (1_000_000, 10_000_000):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(n); e1.record(); torch.cuda.synchronize()
    print(n, e0.elapsed_time(e1),"ms")
