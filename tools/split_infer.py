#!/usr/bin/env python3
"""Inference of one batch as P independent sub-batches on P streams inside one hipGraph (the tail of each kernel of one sub-batch under the
ramp of another's) against the whole batch on one stream.  usage: split_infer.py [model] [batch] [img] [parts ...]"""
import sys, time, torch
sys.path.insert(0, ".")
import lemevit_amd
from lemevit_amd.graph import GraphedStep

name = sys.argv[1] if len(sys.argv) > 1 else "lemevit_base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
img = int(sys.argv[3]) if len(sys.argv) > 3 else 224
parts = [int(a) for a in sys.argv[4:]] or [1, 2, 4]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = lemevit_amd.create_model(name, num_classes=1000).to(dev).eval()
x = torch.randn(B, 3, img, img, device=dev)
ref = None
for P in parts:
    xs = list(x.chunk(P))
    streams = [torch.cuda.Stream() for _ in range(P - 1)]
    outs = [None] * P

    def step():
        cur = torch.cuda.current_stream()
        with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
            for i, s in enumerate(streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs[i + 1] = model(xs[i + 1])
            outs[0] = model(xs[0])
            for s in streams:
                cur.wait_stream(s)

    g = GraphedStep(step, warmup=3)
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            g()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    y = torch.cat([o.float() for o in outs])
    with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
        yseq = torch.cat([model(xi).float() for xi in xs])          # the same sub-batches one after the other on one stream
    print(f"   concurrent vs sequential sub-batches: max |diff| {float((y - yseq).abs().max()):.3e}; max |logit| {float(y.abs().max()):.3f}")
    if ref is None:
        ref = y
    print(f"parts {P}: {best * 1e3:.3f} ms per batch of {B} ({B / best:.0f} img/s)  max |diff| vs first {float((y - ref).abs().max()):.3e}", flush=True)

    def seq_step():              # the same sub-batches one after the other on ONE stream (smaller working set per launch, no concurrency)
        with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
            for i in range(P):
                outs[i] = model(xs[i])

    if P > 1:
        g2 = GraphedStep(seq_step, warmup=2)
        for _ in range(3):
            g2()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g2()
        torch.cuda.synchronize()
        print(f"   sequential on one stream: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
