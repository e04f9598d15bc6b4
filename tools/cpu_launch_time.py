#!/usr/bin/env python3
"""Host time to ISSUE one eager train step (no device sync inside the loop) vs. its device time: how close eager mode is to
being launch-bound.  usage: python tools/cpu_launch_time.py [--sync]   (--sync: with a 1-rank FlatGradSync)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lemevit_amd
dev = torch.device("cuda:0")
if "--pg-first" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = lemevit_amd.create_model("lemevit_base", num_classes=1000, drop_path_rate=0.1).to(dev).train()
opt = lemevit_amd.FlatAdamW(model, lr=1e-4, weight_decay=0.05)
gsync = None
x = torch.randn(128, 3, 224, 224, device=dev); lf = torch.nn.CrossEntropyLoss()
if "--pg-late" in sys.argv:                     # a few steps first: the caching allocator owns its segments before RCCL comes up
    for _ in range(3):
        opt.zero_grad()
        with torch.autocast("cuda", torch.bfloat16):
            lf(model(x), torch.empty((128,), device=dev, dtype=torch.long).random_(1000)).backward()
        opt.step()
    torch.cuda.synchronize()
if "--sync" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from lemevit_amd.dist import attach_flat_grad_sync
    gsync = attach_flat_grad_sync(model, opt, force=True)
x = torch.randn(128, 3, 224, 224, device=dev); lf = torch.nn.CrossEntropyLoss()
def step():
    opt.zero_grad()
    with torch.autocast("cuda", torch.bfloat16):
        out = model(x)
        lf(out, torch.empty((128,), device=dev, dtype=torch.long).random_(1000)).backward()
    if gsync is not None:
        gsync.finish()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time {1e3 * (t1 - t0) / n:.2f} ms/step; wall {1e3 * (t2 - t0) / n:.2f} ms/step (device-bound if wall > issue)")
