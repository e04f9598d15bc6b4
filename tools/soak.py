#!/usr/bin/env python3
"""Soak run (GPU box): the persistent stage kernels under repetition -- N forward passes of one batch, every output compared BIT FOR BIT with the first, the hand-off error
word read at the end (lemevit_amd.ops.check_stage_errors raises on a lost hand-off) -- and K training steps on one fixed batch (finite, falling loss; error word).
usage: soak.py [forward passes] [train steps]"""
import sys, time, torch
sys.path.insert(0, ".")
import lemevit_amd
from lemevit_amd import ops
from lemevit_amd.graph import split_forward
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 300
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
for name, B, res, parts in (("lemevit_base", 128, 224, 4), ("lemevit_base", 128, 224, 1), ("lemevit_tiny", 256, 224, 1), ("lemevit_base", 64, 384, 1)):
    torch.manual_seed(0)
    model = lemevit_amd.create_model(name, num_classes=1000).to(dev).eval()
    x = torch.randn(B, 3, res, res, device=dev)
    outs, first, diff = [], None, 0
    t0 = time.perf_counter()
    with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
        for i in range(NF):
            outs.clear()
            split_forward(model, x, parts, outs)
            o = torch.cat([t.float() for t in outs]) if len(outs) > 1 else outs[0].float()
            if first is None:
                first = o.clone()
            elif not torch.equal(o, first):
                diff += 1
    torch.cuda.synchronize()
    ops.check_stage_errors(f"soak {name}", sync=True)
    print(f"{name} B={B} {res}x{res}, {parts} sub-batch stream(s): {NF} forward passes in {time.perf_counter() - t0:.1f} s, {diff} differ from the first bit for bit, "
          f"stage errors {ops.stage_error_count(reset=False, sync=True)}, finite {bool(torch.isfinite(first).all())}", flush=True)
    del model
torch.manual_seed(0)
model = lemevit_amd.create_model("lemevit_base", num_classes=1000, drop_path_rate=0.1).to(dev).train()
opt = lemevit_amd.FlatAdamW(model, lr=2e-4, eps=1e-8, weight_decay=0.05)
x = torch.randn(128, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (128,), device=dev)
losses = []
t0 = time.perf_counter()
for i in range(NT):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(model(x), y)
    loss.backward(); opt.step()
    if i % 10 == 0 or i == NT - 1:
        losses.append(float(loss.detach()))
torch.cuda.synchronize()
ops.check_stage_errors("soak train", sync=True)
print(f"lemevit_base train, one fixed batch of 128, {NT} steps in {time.perf_counter() - t0:.1f} s: loss {losses[0]:.3f} -> {losses[-1]:.3f} "
      f"(every 10th: {' '.join(f'{v:.2f}' for v in losses)}), all finite {all(v == v and abs(v) < 1e9 for v in losses)}")
