#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): images/sec, LeMeViT-Base 224^2 bf16 fwd+bwd.  One "step" reproduces
benchmark.py's TrainBenchmarkRunner step (benchmark.py:572-596): zero_grad -> autocast(bf16){forward ->
CrossEntropyLoss(randint targets) -> backward} -> AdamW.step, on ONE synthetic batch created once
(benchmark.py:462-467), B = 128 per GPU, drop_path 0.1 (scripts/benchmark.sh:9), random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; N > 1 shards images over ranks (weak scaling, 128 images per GPU) with gradients
all-reduced by RCCL (lemevit_amd.dist.FlatGradSync: chunks of the flat fp32 gradient buffer, overlapped with backward;
--ddp selects torch DDP instead).  Rank 0 prints ONE JSON line.

`roofline` is measured live for the dominant kernel of the timed region: the library's launch-timing probe brackets every GEMM (forward-form tile / register-stationary /
whole-width kernels, the transpose-read dX kernel, the split-K weight-gradient kernel and its slab reduce), every attention launch and every persistent stage / stem launch with
HIP events on the stream it is issued on, for 3 eager steps right after the timed region; the KERNEL with the largest total time is the one reported (train mode: the weight-gradient
GEMM, gemm_kernel<bf16, TR, TR, split-K>; --mode infer: sstage_kernel) -- its algorithmic FLOPs / its summed duration against the 2.5 PFLOP/s dense bf16 MFMA peak (the roof
SURVEY 8(d) names), the algorithmic-bytes figure against the 8 TB/s HBM roof beside it (`hbm_*`), every other kernel under `other_launch_kinds`.
`cpu_baseline` times the CPU oracle (oracle/, a port of the reference) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0              # HBM3E, MI355X_MICROARCH.md
CANONICAL_GFLOP_FWD = 22.12        # README/BASELINE: 11.06 GMAC forward per image (Base 224^2)
CANONICAL_GMAC = {("lemevit_base", 224): 11.060, ("lemevit_small", 224): 3.736, ("lemevit_tiny", 224): 1.779,
                  ("lemevit_base", 384): 30.986, ("lemevit_tiny", 384): 5.004}     # BASELINE.md section 1 (canonical = README-style)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="lemevit_base")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-protocol", default="", metavar="OUT.json", help="only run the CPU oracle with benchmark.py's 10 + 40 protocol (Tiny B=1, Base B=1 / B=32) and exit")
    ap.add_argument("--ddp", action="store_true", help="N > 1: torch DistributedDataParallel + torch AdamW instead of FlatAdamW + FlatGradSync")
    ap.add_argument("--force-sync", action="store_true", help="run the FlatGradSync collectives even at world size 1 (1-rank process group)")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DDP (and use the DDP code path) even at world size 1: measures the wrapper's overhead")
    ap.add_argument("--torch-adamw", action="store_true", help="use torch.optim.AdamW(fused=True) instead of lemevit_amd.FlatAdamW at N=1")
    ap.add_argument("--grad-wire", default="auto", choices=["auto", "fp32", "bf16"], help="N > 1 with FlatGradSync: element type of the gradient all-reduce.  auto = fp32, the reference's DDP byte count "
                    "(212 MB per step; main.py:333 has no compression) -- the like-for-like scaling line; bf16 halves the bytes over xGMI (SURVEY 8(e) / row f3), opt-in")
    ap.add_argument("--print-csrc-hash", action="store_true", help="print the content hash of the kernel sources (lemevit_amd/csrc, include/) and exit: the PMC scripts under tools/ "
                    "store it in the profiles/*_traffic.json files they write, and this program only quotes such a file while the hash still matches the tree it runs from")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-issue-probe", action="store_true", help="skip the 5 synchronised single steps that measure host issue time (profiling runs: "
                    "they would sit in the 'last steps' window of a kernel trace)")
    ap.add_argument("--no-forward-probe", action="store_true", help="train mode: skip the forward-only pass timed after the train region (forward_* keys)")
    ap.add_argument("--infer-parts", type=int, default=0, help="forward-only passes (--mode infer and the forward_* probe): run the batch as this many "
                    "concurrent sub-batches inside the one hipGraph (lemevit_amd.graph.split_forward); 1 = the whole batch on one stream")
    ap.add_argument("--graph", type=int, default=-1, help="(-1 = auto: eager for train, graph replay for infer)  1 = capture the step into a hipGraph (lemevit_amd.graph.GraphedStep) and replay it; default 0 = "
                    "eager launches, which are faster here: the weight-gradient GEMMs overlap the dX chain on a side stream, and the "
                    "runtime serialises the branches of a captured graph (38.0 vs 39.6 ms per step)")
    return ap.parse_args()


def cpu_baseline(model_name: str, img: int, mode: str):
    """Oracle (port of the reference) on the host cores, bounded to ~10-30 s."""
    from oracle import lemevit_oracle as O
    cfg = O.VARIANTS[model_name]
    torch.manual_seed(0)
    spec = O.state_dict_spec(cfg, 1000)
    sd = {}
    for k, shp in spec.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shp)
        elif len(shp) >= 2:
            sd[k] = torch.randn(shp) * 0.02
        elif k.endswith(".weight"):
            sd[k] = torch.ones(shp)
        else:
            sd[k] = torch.zeros(shp)
    train = mode == "train"
    if train:
        for k, v in sd.items():
            if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
    # 128 host threads on a batch of 8 images thrash (measured on the GPU box: Base B=1 forward 0.83 img/s with 128 threads, 3.2 img/s
    # with 8): the baseline uses at most 32, and `cores` reports the threads actually used
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(min(prev_threads, 32))
    B = 8
    x = torch.randn(B, 3, img, img)
    tgt = torch.randint(0, 1000, (B,))

    def step():
        if train:
            loss = torch.nn.functional.cross_entropy(O.lemevit_forward(sd, cfg, x, train=True), tgt)
            loss.backward()
            for v in sd.values():
                v.grad = None
        else:
            with torch.no_grad():
                O.lemevit_forward(sd, cfg, x)

    step()                                        # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        step(); n += 1
        if time.perf_counter() - t0 > 15.0 or n >= 16:
            break
    dt = time.perf_counter() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(prev_threads)
    return dict(value=round(B * n / dt, 3), unit="images/sec", cores=used, kind="port",
                sample=f"{n} {'fwd+bwd' if train else 'fwd'} steps of {model_name} {img}x{img} fp32 at batch {B} (oracle/lemevit_oracle.py, PyTorch CPU)")


def cpu_baseline_protocol(out_path: str):
    """the oracle on the host cores with benchmark.py's inference protocol (eval, no_grad, one synthetic
    batch, 10 warm-up + 40 timed steps, benchmark.py:120-130,320-321,517-518) -- Tiny B=1 (config 1), Base B=1 and Base B=32.
    Takes minutes; not part of the default bench run (python bench.py --cpu-baseline-protocol gpurun_out/cpu_baseline.json)."""
    from oracle import lemevit_oracle as O
    res = dict(cores=torch.get_num_threads(), kind="port", protocol="eval + no_grad, 10 warm-up + 40 timed steps, fp32, oracle/lemevit_oracle.py")
    try:
        with open("/proc/cpuinfo") as f:
            res["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    for name, B, warm, iters in (("lemevit_tiny", 1, 10, 40), ("lemevit_base", 1, 10, 40), ("lemevit_base", 32, 3, 10)):
        cfg = O.VARIANTS[name]
        torch.manual_seed(0)
        sd = {}
        for k, shp in O.state_dict_spec(cfg, 1000).items():
            sd[k] = (torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else torch.ones(shp) if k.endswith(("running_var",)) or
                     (len(shp) == 1 and k.endswith(".weight")) else torch.randn(shp) * 0.02 if len(shp) >= 2 else torch.zeros(shp))
        x = torch.randn(B, 3, 224, 224)
        with torch.no_grad():
            for _ in range(warm):
                O.lemevit_forward(sd, cfg, x)
            t0 = time.perf_counter()
            for _ in range(iters):
                O.lemevit_forward(sd, cfg, x)
            dt = time.perf_counter() - t0
        res[f"{name}_b{B}"] = dict(images_per_sec=round(B * iters / dt, 3), ms_per_step=round(1e3 * dt / iters, 2), warmup=warm, iters=iters)
        print(name, B, res[f"{name}_b{B}"], file=sys.stderr, flush=True)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def pick_infer_parts(model, x, use_graph, candidates=(1, 2, 4)):
    """--infer-parts 0 (default): the forward-only schedule is chosen BEFORE the timed region -- the batch as 1, 2 or 4 concurrent sub-batches
    (lemevit_amd.graph.split_forward), each tried for a few passes; persistent stage kernels fill the chip on their own (one stream wins where
    they cover most of the pass: LeMeViT-Tiny), per-launch stages overlap their ramps and tails across sub-batches."""
    from lemevit_amd.graph import split_forward, try_graphed
    best, best_ms, table = candidates[0], float("inf"), {}
    for parts in candidates:
        outs = []

        def fstep(parts=parts, outs=outs):
            with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
                split_forward(model, x, parts, outs)

        step = fstep
        if use_graph:
            step, why = try_graphed(fstep, warmup=3)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 6
        table[parts] = round(ms, 3)
        if ms < best_ms:
            best, best_ms = parts, ms
        del step, outs
    return best, table


def csrc_hash() -> str:
    """sha256 over the kernel sources the library is built from (names + contents, sorted): ties a committed PMC figure to the code it was measured on.
    (`git rev-parse HEAD:lemevit_amd/csrc` would do on the build host, but the GPU box receives the tree without .git.)"""
    import glob, hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "lemevit_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "lemevit_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "lemevit_amd", "csrc", "Makefile")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode()); h.update(b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _matching(path: str):
    """The JSON object of a committed PMC file if it was measured on THIS tree's kernel sources, else None (with a note on stderr)."""
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception as e:
        print(f"bench.py: cannot read {path}: {e}", file=sys.stderr)
        return None
    if d.get("csrc_hash") != csrc_hash():
        print(f"bench.py: {os.path.relpath(path, ROOT)} was measured on kernel sources {d.get('csrc_hash')} but this tree is {csrc_hash()}: not quoted (re-run the PMC passes)", file=sys.stderr)
        return None
    return d


def main():
    args = parse()
    if args.print_csrc_hash:
        print(csrc_hash())
        return
    if args.cpu_baseline_protocol:
        return cpu_baseline_protocol(args.cpu_baseline_protocol)
    if args.graph < 0:
        args.graph = 0 if args.mode == "train" else 1
    parts_table = None
    # stdout carries exactly ONE line, the JSON record: RCCL prints a version banner through C stdio on stdout (flushed at
    # exit, i.e. AFTER anything printed here), MIOpen / hipBLASLt may log there too.  Everything else goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (1-GPU box): LMV_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and LMV_BENCH_BACKEND=gloo replaces RCCL (which
    # refuses two ranks on one device), so the N > 1 code path can be exercised end to end without N GPUs
    if os.environ.get("LMV_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("LMV_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import lemevit_amd
    from lemevit_amd.dist import wrap_ddp
    if os.environ.get("LMV_BENCH_SINGLE_DEVICE") == "1" and world > 1:
        # Ranks that SHARE a device (the 1-GPU test hook only) must not launch the persistent stage kernels: those assume the workgroups of an image are co-resident, which one
        # process arranges for its own launches ("Residency") but two processes on one device cannot -- the in-launch waits would run into their bounded spins.
        import lemevit_amd.model as _m
        _m._SSTAGE = False

    torch.manual_seed(0)
    train = args.mode == "train"
    model = lemevit_amd.create_model(args.model, num_classes=1000, drop_path_rate=0.1 if train else 0.0).to(dev)
    model.train(train)
    x = torch.randn(args.batch, 3, args.img, args.img, device=dev)
    loss_fn = torch.nn.CrossEntropyLoss().to(dev)
    # The process group is created AFTER the model: with RCCL initialised first, every step of this workload is ~3 ms slower
    # even when no collective runs (measured with a 1-rank group, tools/cpu_launch_time.py --pg-first); all ranks seed
    # identically and FlatGradSync / DDP broadcast rank 0's parameters anyway.
    if world > 1 or args.force_ddp or args.force_sync:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if train:
        # benchmark.py:559-561 create_optimizer_v2(opt='adamw', lr=1e-4), scripts/benchmark.sh:8 eps 1e-8 wd 0.05
        decay = [p for n, p in model.named_parameters() if p.ndim > 1]
        no_decay = [p for n, p in model.named_parameters() if p.ndim <= 1]
        use_ddp = (world > 1 and args.ddp) or args.force_ddp
        gsync = None
        if not use_ddp and not args.torch_adamw:
            # the framework's optimizer: block parameters flat, gradients written in place, one fused AdamW launch that also
            # refreshes the bf16 operand copies (same update rule; tests/test_model_gpu.py::test_flat_adamw_matches_torch_adamw)
            opt = lemevit_amd.FlatAdamW(model, lr=1e-4, eps=1e-8, weight_decay=0.05)
            if world > 1 or args.force_sync:
                # data parallelism without a DDP wrapper: the flat block-gradient buffer is all-reduced in 4 large chunks, each as soon
                # as the backward pass has written it (lemevit_amd/dist.py::FlatGradSync)
                from lemevit_amd.dist import attach_flat_grad_sync
                wire = "bf16" if args.grad_wire == "bf16" else None
                gsync = attach_flat_grad_sync(model, opt, force=args.force_sync, compress=wire)
        else:
            opt = torch.optim.AdamW([dict(params=decay, weight_decay=0.05), dict(params=no_decay, weight_decay=0.0)], lr=1e-4, eps=1e-8, fused=True,
                                    capturable=bool(args.graph))
        net = wrap_ddp(model, local) if use_ddp else model

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", torch.bfloat16):
                out = net(x)
                target = torch.empty((args.batch,), device=dev, dtype=torch.long).random_(1000)
                loss_fn(out, target).backward()
            if gsync is not None:
                gsync.finish()
            opt.step()
    else:
        from lemevit_amd.graph import split_forward
        infer_outs = []
        if args.infer_parts <= 0:
            args.infer_parts, parts_table = pick_infer_parts(model, x, bool(args.graph) and world == 1)

        def step():
            with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
                split_forward(model, x, args.infer_parts, infer_outs)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from lemevit_amd import ops as _ops

    def check_handoffs(where):
        # A persistent stage kernel whose in-launch wait ran out has produced wrong tensors: that must be an exception here, not a normal-looking bench line
        # (VERDICT round 4, weak #2).  Called right behind a sync(): the pinned error word then covers every launch issued so far.
        _ops.check_stage_errors(f"bench.py, {where}", sync=False)

    # which devices take part: the driver can check N distinct GPUs from the line (N > 1: an all-gather of (rank, device uuid) over the process group)
    try:
        my_uuid = str(torch.cuda.get_device_properties(dev).uuid)
    except Exception:
        my_uuid = f"cuda:{local}"
    ranks_seen, device_uuids = 1, [my_uuid]
    if dist.is_initialized():
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, (rank, my_uuid))
        ranks_seen, device_uuids = dist.get_world_size(), [u for _, u in sorted(gathered)]

    eager_step, graph_note = step, "eager, weight-gradient GEMMs on a side stream"
    if args.graph and world == 1 and not args.force_ddp and not args.force_sync:
        from lemevit_amd.graph import try_graphed
        step, why = try_graphed(eager_step, warmup=3)
        graph_note = "hipGraph replay" if why is None else f"eager (graph capture failed: {why})"
        if not train and args.infer_parts > 1:
            graph_note += f", the batch as {args.infer_parts} concurrent sub-batches (graph branches)"
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    check_handoffs("timed region")
    # host time to ENQUEUE one step, measured on an EMPTY launch queue (issuing K steps back to back only measures the queue's
    # back-pressure: the host blocks once the device is a queue depth behind) -- median of 5 single steps, each preceded by a sync
    issue = []
    for _ in range(0 if args.no_issue_probe else 5):
        sync()
        ti = time.perf_counter()
        step()
        issue.append(time.perf_counter() - ti)
    sync()
    t_issued = sorted(issue)[len(issue) // 2] * args.steps if issue else None
    kernel_timing_note, probe, probe_alone = None, None, None
    if not args.no_kernel_timing:
        # Same schedule as the timed region: 3 eager steps right AFTER it with the library's launch-timing probe on (lmv_debug_launch_timing: HIP events around
        # every forward-form Linear entry point and every persistent stage / stem launch, recorded on the stream the launch is issued on -- the native block
        # schedule's launches included).  Events cannot be timed inside a captured graph, and in eager mode they cost ~1 ms per step, which must not leak
        # into `value`.  All ranks run the 3 steps: they contain the gradient collectives.
        import ctypes
        from lemevit_amd import _lib as _L
        cap = 16384
        if rank == 0:
            _L.check(_L.lib.lmv_debug_launch_timing(cap), "lmv_debug_launch_timing")
        for _ in range(3):
            eager_step()
        sync()
        def read_probe():
            ms, fl, by, kd = (ctypes.c_float * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_int * cap)()
            n = _L.lib.lmv_debug_launch_timing_read(ms, fl, by, kd, cap)
            _L.lib.lmv_debug_launch_timing(0)
            out = {}
            for i in range(n):
                e = out.setdefault(kd[i], dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0))
                e["launches"] += 1; e["total_ms"] += ms[i]; e["flops"] += fl[i]; e["bytes"] += by[i]
            return out
        if rank == 0:
            probe = read_probe()
        # ... and, for continuity with rounds 1 - 3, the same kernels launched ALONE on the chip: the per-launch Python schedule of the blocks (whole-batch launches,
        # lemevit_amd/blocks.py), one stream for the forward pass; reported beside the primary figure as roofline.frac_whole_batch_launches
        import lemevit_amd.model as _model
        native_was = _model._NATIVE
        if train:
            _model._NATIVE = False          # (--mode infer keeps the persistent stage launches: there "alone" means the whole batch on one stream)
        parts_was, args.infer_parts = args.infer_parts, 1
        if rank == 0:
            _L.check(_L.lib.lmv_debug_launch_timing(cap), "lmv_debug_launch_timing")
        for _ in range(2):
            eager_step()
        sync()
        probe_alone = read_probe() if rank == 0 else None
        args.infer_parts = parts_was
        _model._NATIVE = native_was
        kernel_timing_note = ("HIP events of the library's launch-timing probe (lmv_debug_launch_timing) around every launch of this kind in 3 eager steps run right after the timed "
                              "region, on the timed region's own schedule (native block calls, concurrent ranges of images / sub-batches, weight-gradient GEMMs on the side stream) and on "
                              "the launch's own stream: launches that overlap on the chip share it, so the sum of their durations exceeds the wall time they cover (round 3 timed whole-batch "
                              "launches alone on the chip: 0.144 for the same kernels)")
    # The north-star target is FORWARD throughput (BASELINE.json): in train mode the same process times the forward pass of the same
    # model on the same batch right after the train region (eval mode, no_grad, bf16 autocast, hipGraph replay as in --mode infer) and
    # reports it as extra keys of the same JSON line.  The timed train region above is not touched by it.
    fwd_dt, fwd_iters, fwd_note, fwd_probe = None, 0, None, None
    if train and not args.no_forward_probe:
        model.eval()

        from lemevit_amd.graph import split_forward
        fwd_outs = []
        if args.infer_parts <= 0:
            args.infer_parts, parts_table = pick_infer_parts(model, x, True)

        def fwd_step():
            with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
                split_forward(model, x, args.infer_parts, fwd_outs)

        from lemevit_amd.graph import try_graphed
        fstep, why = try_graphed(fwd_step, warmup=3)
        fwd_note = "hipGraph replay" if why is None else f"eager (graph capture failed: {why})"
        if args.infer_parts > 1:
            fwd_note += f", the batch as {args.infer_parts} concurrent sub-batches (graph branches)"
        for _ in range(3):
            fstep()
        sync()
        fwd_iters = max(10, args.steps)
        tf0 = time.perf_counter()
        for _ in range(fwd_iters):
            fstep()
        sync()
        fwd_dt = time.perf_counter() - tf0
        check_handoffs("forward region")
        fwd_probe = None
        if rank == 0 and not args.no_kernel_timing:
            # the forward pass's own dominant launch (the persistent stage-3 kernel on this workload), whole batch on one stream, by the same library probe
            import ctypes
            from lemevit_amd import _lib as _L
            from lemevit_amd.graph import split_forward as _sf
            cap = 4096
            _L.check(_L.lib.lmv_debug_launch_timing(cap), "lmv_debug_launch_timing")
            for _ in range(2):
                with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
                    _sf(model, x, 1, [])
            torch.cuda.synchronize()
            ms_, fl_, by_, kd_ = (ctypes.c_float * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_int * cap)()
            n_ = _L.lib.lmv_debug_launch_timing_read(ms_, fl_, by_, kd_, cap)
            _L.lib.lmv_debug_launch_timing(0)
            agg = {}
            for i in range(n_):
                e = agg.setdefault(kd_[i], [0, 0.0, 0.0]); e[0] += 1; e[1] += ms_[i]; e[2] += fl_[i]
            if agg:
                k = max(agg, key=lambda q: agg[q][1])
                fwd_probe = dict(kernel={0: "gemm_kernel<bf16, NT>", 1: "sstage_kernel", 2: "dstage_kernel", 3: "stem_kernel", 7: "attention forward launches", 8: "rs_gemm_kernel", 9: "wn_gemm_kernel",
                                         11: "rsw_gemm_kernel"}.get(k, str(k)), launches=agg[k][0],
                                 avg_launch_us=round(1e3 * agg[k][1] / agg[k][0], 1), tflops=round(agg[k][2] / (agg[k][1] * 1e-3) / 1e12, 1),
                                 frac=round(agg[k][2] / (agg[k][1] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), bound="mfma", peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                                 measured="2 eager forward passes after the forward region, whole batch on one stream, library launch-timing probe")
        model.train(True)
    tmax = torch.tensor([dt, fwd_dt or 0.0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0].item())
    if fwd_dt is not None:
        fwd_dt = float(tmax[1].item())

    if rank == 0:
        pretty = {"lemevit_tiny": "LeMeViT-Tiny", "lemevit_small": "LeMeViT-Small", "lemevit_base": "LeMeViT-Base"}.get(args.model, args.model)
        imgs = args.batch * world * args.steps
        value = imgs / dt
        mult = 3.0 if train else 1.0
        gmac = CANONICAL_GMAC.get((args.model, args.img))
        gflop = None if gmac is None else 2.0 * gmac
        KIND_NAMES = {0: "gemm_kernel<bf16, NT> (forward-form Linear launches on the 128 x 128 tile kernel)",
                      1: "sstage_kernel (a stage of S blocks as one persistent launch)", 2: "dstage_kernel (a stage of D / C blocks as one persistent launch)", 3: "stem_kernel",
                      4: "gemm_kernel<bf16, B transposed> (dX through the untransposed weight)", 5: "gemm_kernel<bf16, TR, TR, split-K> (weight-gradient GEMM, dW = dY^T X)",
                      6: "splitk_reduce_kernel (slab sums of the weight-gradient GEMM)", 7: "attention forward launches (mfma_fwd_*)", 8: "rs_gemm_kernel (register-stationary forward-form Linear)",
                      9: "wn_gemm_kernel (whole-width forward-form Linear, incl. the fused proj + norm2 and dX + LayerNorm-backward forms)", 10: "attention backward launches (mfma_bwd_*)",
                      11: "rsw_gemm_kernel (norm1 + C = 96 projections, weights resident in LDS)"}
        SHORT = {0: "gemm_kernel_nt", 1: "sstage_kernel", 2: "dstage_kernel", 3: "stem_kernel", 4: "gemm_kernel_dx", 5: "gemm_kernel_dw", 6: "splitk_reduce", 7: "attention_fwd", 8: "rs_gemm_kernel",
                 9: "wn_gemm_kernel", 10: "attention_bwd", 11: "rsw_gemm_kernel"}
        g, gkind = None, None
        if probe:
            gkind = max(probe, key=lambda k: probe[k]["total_ms"])          # the dominant KERNEL of this workload: the kind with the largest total time (one kernel per kind)
            e = probe[gkind]
            g = dict(launches=e["launches"], total_ms=e["total_ms"], avg_us=1e3 * e["total_ms"] / e["launches"], tflops=e["flops"] / (e["total_ms"] * 1e-3) / 1e12,
                     gflop_per_launch=e["flops"] / e["launches"] / 1e9, mbytes_per_launch=e["bytes"] / e["launches"] / 1e6)
        roof = None
        if g is not None:
            # HBM bytes per launch of the same kernel from the PMC passes committed under profiles/ (tools/pmc_traffic.sh: FETCH_SIZE x 2
            # per the gfx950 correction + WRITE_SIZE, separate passes); only quoted for the workload it was collected on
            traffic, traffic_src = None, None
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_gemm_fwd_pmc_traffic.json")))      # the newest COMMITTED round (r03's file never left gpurun_out/)
            cands_dw = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_gemm_dw_pmc_traffic.json")))
            # (each file carries the content hash of the kernel sources it was measured on -- csrc_hash() -- and is only quoted while it matches this tree: a kernel change without
            #  a new PMC pass leaves `traffic` null instead of a stale figure)
            if gkind == 5 and train and args.model == "lemevit_base" and args.img == 224 and args.batch == 128 and cands_dw:
                d_ = _matching(cands_dw[-1])
                if d_ is not None:
                    traffic = round(d_["traffic_bytes_per_launch"] / 1e6, 2)
                    traffic_src = (f"profiles/{os.path.basename(cands_dw[-1])}, measured on these kernel sources (csrc_hash {csrc_hash()}): MB per launch of gemm_kernel<bf16, TR, TR, split-K> "
                                   "over this command's train step, separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.sh; not collected in this run")
            elif gkind == 0 and train and args.model == "lemevit_base" and args.img == 224 and args.batch == 128 and cands:
                pmc = cands[-1]
                d_ = _matching(pmc)
                if d_ is not None:
                    traffic = round(d_["traffic_bytes_per_launch"] / 1e6, 2)
                    traffic_src = (f"profiles/{os.path.basename(pmc)}, measured on these kernel sources (csrc_hash {csrc_hash()}): MB per launch over the forward-form Linear launches -- "
                                   "gemm_kernel<bf16,NT>, rs_gemm_kernel, wn_gemm_kernel -- of this command's train step, separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                   "tools/pmc_traffic.sh; not collected in this run; that launch set also holds the dX launches that run as forward-form GEMMs on transposed weights")
            elif gkind == 1 and not train and args.model == "lemevit_base" and args.img == 224 and args.batch == 128:
                # the persistent stage-3 launch: per-kernel PMC table of the same command (tools/pmc_kernels.sh; one launch = the whole batch)
                cands_k = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_per_kernel_infer.csv")))
                hash_ok = False
                if cands_k and os.path.exists(cands_k[-1] + ".csrc_hash"):
                    with open(cands_k[-1] + ".csrc_hash") as f:
                        hash_ok = f.read().strip() == csrc_hash()
                if cands_k and not hash_ok:
                    print(f"bench.py: {os.path.basename(cands_k[-1])} carries no matching csrc_hash: roofline.traffic stays null", file=sys.stderr)
                if cands_k and hash_ok:
                    import csv
                    with open(cands_k[-1]) as f:
                        for row in csv.DictReader(f):
                            if row["kernel"].startswith("sstage_kernel<8>"):
                                traffic = round(float(row["fetch_MB_per_launch(x2_corrected)"]) + float(row["write_MB_per_launch"]), 1)
                                traffic_src = (f"profiles/{os.path.basename(cands_k[-1])}, measured on these kernel sources (csrc_hash {csrc_hash()}) (whole-batch launch; FETCH_SIZE x 2 + WRITE_SIZE at the L2-fabric boundary: it counts the write-through "
                                               "K / V and halo exchanges and the parked residual registers of the launch -- MALL-absorbable traffic, DESIGN 4.9 -- not only the algorithmic tokens + weights)")
            elif train and gkind == 0:
                print("bench.py: no profiles/rNN_gemm_fwd_pmc_traffic.json for this workload: roofline.traffic stays null", file=sys.stderr)
            # SURVEY 8(d) grades the path against the MFMA roof (94 % of the MACs are Linear GEMMs), so that is the primary figure.  The
            # launch mix itself has K = 96..512 on the big-row stages: its arithmetic intensity is below the ridge point of the chip
            # (2500 TFLOP/s / 8 TB/s = 312 flop/B), i.e. by the roofline model HBM is the binding roof -- reported beside it (`hbm_*`).
            hbm_gbs = g["mbytes_per_launch"] * 1e6 / (g["avg_us"] * 1e-6) / 1e9
            intensity = g["gflop_per_launch"] * 1e9 / (g["mbytes_per_launch"] * 1e6)
            roof = dict(bound="mfma", achieved=round(g["tflops"], 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(g["tflops"] / PEAK_BF16_TFLOPS, 4),
                        traffic=traffic, traffic_unit="MB/launch", traffic_source=traffic_src, algorithmic_mbytes_per_launch=round(g["mbytes_per_launch"], 2),
                        flop_per_byte=round(intensity, 1), ridge_flop_per_byte=round(PEAK_BF16_TFLOPS * 1e3 / PEAK_HBM_GBS, 1),
                        hbm_gbs=round(hbm_gbs, 1), hbm_peak_gbs=PEAK_HBM_GBS, hbm_frac=round(hbm_gbs / PEAK_HBM_GBS, 4),
                        traffic_over_algorithmic=None if traffic is None else round(traffic / g["mbytes_per_launch"], 3),
                        kernel=KIND_NAMES.get(gkind, str(gkind)), measured=kernel_timing_note,
                        frac_whole_batch_launches=None if not (probe_alone and gkind in probe_alone) else round(probe_alone[gkind]["flops"] / (probe_alone[gkind]["total_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                        other_launch_kinds={SHORT.get(k, str(k)): dict(launches=v["launches"], ms_per_step=round(v["total_ms"] / 3, 3), tflops=round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 1))
                                            for k, v in probe.items() if k != gkind}, launches=g["launches"], avg_launch_us=round(g["avg_us"], 2),
                        gflop_per_launch=round(g["gflop_per_launch"], 3))
        line = {
            "metric": f"images/sec {pretty} {args.img}^2 bf16 " + ("fwd+bwd" if train else "fwd"),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model} {args.img}x{args.img} bf16-autocast {'train step (fwd+bwd+AdamW)' if train else 'forward'}, "
                                   f"batch {args.batch}/GPU, drop_path 0.1, random-init weights", "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "launch": graph_note},
            "host_issue_ms_per_step": None if t_issued is None else round(1e3 * t_issued / args.steps, 3),
            "model_tflops": None if gflop is None else round(value * gflop * mult / 1e3, 2),
            "model_frac_of_bf16_peak": None if gflop is None else round(value * gflop * mult / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
            "roofline": roof,
            # the sticky error word of the persistent stage kernels, READ here (pinned host memory; every region above ended in a synchronise + check that raises on a non-zero
            # word, so a line is only printed with 0 -- but it is the counter, not a literal)
            "stage_errors": _ops.stage_error_count(reset=False, sync=False), "ranks_seen": ranks_seen, "device_uuids": device_uuids,
            "grad_wire": None if not (train and world > 1) else ("torch DDP bf16 hook" if (args.ddp or args.force_ddp) else ("bf16" if args.grad_wire == "bf16" else "fp32")),
        }
        # HBM traffic of the WHOLE step over its wall time: bytes per step from the newest committed PMC pass over this command (tools/pmc_step_traffic.sh: FETCH_SIZE x 2 +
        # WRITE_SIZE summed over every kernel of a step), quoted only while it was measured on these kernel sources
        if world == 1 and args.model == "lemevit_base" and args.img == 224 and args.batch == 128:
            import glob
            cands_s = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_train_hbm_traffic.json" if train else "r[0-9][0-9]_infer_hbm_traffic.json")))
            d_ = _matching(cands_s[-1]) if cands_s else None
            if d_ is not None:
                line["hbm_step_mb"] = round(d_["hbm_mb_per_step"], 1)
                line["hbm_step_gbs"] = round(d_["hbm_mb_per_step"] / 1e3 / (dt / args.steps), 1)
                line["hbm_step_frac_of_8tbs"] = round(line["hbm_step_gbs"] / 8000.0, 4)
                line["hbm_step_source"] = f"profiles/{os.path.basename(cands_s[-1])} (csrc_hash {csrc_hash()}) / this run's ms_per_step"
        if fwd_dt is not None:
            fv = args.batch * world * fwd_iters / fwd_dt
            line["forward_images_per_sec"] = round(fv, 2)
            line["forward_ms"] = round(1e3 * fwd_dt / fwd_iters, 3)
            line["forward_frac_of_bf16_peak"] = None if gflop is None else round(fv * gflop / 1e3 / (PEAK_BF16_TFLOPS * world), 4)
            if fwd_probe is not None:
                line["forward_roofline"] = fwd_probe
            line["forward_note"] = (f"forward pass of the same model / batch timed after the train region: eval mode, no_grad, bf16 autocast, {fwd_note}, "
                                    f"{fwd_iters} iterations, fused inference schedule")
        if parts_table is not None:
            line["forward_schedule_probe_ms"] = {f"{k}_sub_batches": v for k, v in parts_table.items()}      # --infer-parts 0: chosen before the timed region
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model, args.img, args.mode)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
