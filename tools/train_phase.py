#!/usr/bin/env python3
"""Forward and backward phases of the Base train step timed separately (device synchronised between them), per number of concurrent image
ranges of the forward pass (model.TRAIN_PARTS).  usage: train_phase.py [parts ...]"""
import sys, time, torch
sys.path.insert(0, ".")
import lemevit_amd, lemevit_amd.model as M
from lemevit_amd.optim import FlatAdamW

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = lemevit_amd.create_model("lemevit_base", num_classes=1000, drop_path_rate=0.1).to(dev).train()
opt = FlatAdamW(model, lr=1e-4, eps=1e-8, weight_decay=0.05)
x = torch.randn(128, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (128,), device=dev)
lossf = torch.nn.CrossEntropyLoss()
for parts in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    M.TRAIN_PARTS = parts
    tf = tb = 0.0
    n = 0
    for it in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.autocast("cuda", torch.bfloat16):
            loss = lossf(model(x), y)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step(); opt.zero_grad(set_to_none=False)
        if it >= 4:
            tf += t1 - t0; tb += t2 - t1; n += 1
    print(f"parts {parts}: forward {tf / n * 1e3:.2f} ms  backward {tb / n * 1e3:.2f} ms", flush=True)
    # device time of the forward pass with the host AHEAD of the device (a ~20 ms stack of matrix products is queued first)
    big = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    acc = 0.0
    for it in range(6):
        torch.cuda.synchronize()
        for _ in range(40):
            big @ big
        e0.record()
        with torch.autocast("cuda", torch.bfloat16):
            loss = lossf(model(x), y)
        e1.record()
        loss.backward(); opt.step(); opt.zero_grad(set_to_none=False)
        torch.cuda.synchronize()
        if it >= 2:
            acc += e0.elapsed_time(e1)
    print(f"   forward, device time with the host ahead: {acc / 4:.2f} ms", flush=True)
