"""Data parallelism for the LeMeViT hot path: one process per GPU, images sharded over ranks, ONE exchange per step --
the gradient all-reduce (reference: main.py:333 NativeDDP, launched by scripts/train.sh:3-8).

On ROCm the "nccl" backend IS RCCL; on an MI355X node the transport is xGMI (fully connected, 7 links x ~153 GB/s per
GPU).  LeMeViT-Base has 53.1 M parameters = 212 MB of fp32 gradients: compressing the buckets to bf16 halves the bytes
on the per-link-bound ring, and a few LARGE buckets (default 100 MB here, vs torch's 25 MB) suit point-to-point xGMI better
than many small ones.  Everything runs on torch.distributed so the same code is testable with gloo on CPU.
"""
import contextlib
import os
from typing import Optional

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Process group from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        use_gpu = torch.cuda.is_available()
        backend = backend or ("nccl" if use_gpu else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def bf16_compress_hook(process_group, bucket: dist.GradBucket) -> torch.futures.Future[torch.Tensor]:
    """DDP comm hook: all-reduce the bucket as bf16 (half the xGMI bytes), average, decompress into the fp32 bucket."""
    group = process_group if process_group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    buf = bucket.buffer()
    comp = buf.to(torch.bfloat16).div_(world)
    fut = dist.all_reduce(comp, group=group, async_op=True).get_future()

    def decompress(f):
        buf.copy_(f.value()[0])
        return buf

    return fut.then(decompress)


def wrap_ddp(model: torch.nn.Module, local_rank: Optional[int] = None, bf16_grads: bool = True, bucket_cap_mb: int = 100) -> DDP:
    """DistributedDataParallel with xGMI-sized buckets and (optionally) bf16 gradient compression.
    broadcast_buffers=True keeps the reference's per-forward BatchNorm-statistics broadcast (main.py:333)."""
    on_gpu = next(model.parameters()).is_cuda
    ddp = DDP(model, device_ids=[local_rank] if on_gpu else None, broadcast_buffers=True, find_unused_parameters=False,
              bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
    if bf16_grads:
        ddp.register_comm_hook(None, bf16_compress_hook)
    return ddp


def shard_batch(global_batch: int, rank: int, world: int) -> range:
    """Indices of the global batch owned by `rank` (even split; the reference requires divisibility, main.py:~400)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return range(rank * per, (rank + 1) * per)


def all_reduce_mean(t: torch.Tensor) -> torch.Tensor:
    """timm.utils.reduce_tensor (engine.py:137,220-222): clone -> all_reduce(SUM) -> / world."""
    if not dist.is_initialized():
        return t
    rt = t.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    return rt / dist.get_world_size()


class FlatGradSync:
    """Gradient exchange for a model trained with ``lemevit_amd.FlatAdamW``: no DistributedDataParallel wrapper.

    DDP copies every parameter's gradient into its bucket (561 small copies per step for LeMeViT-Base: +1.9 ms, measured)
    and cannot see gradients that the block backward writes in place.  Here the block gradients already live in ONE flat
    fp32 buffer in forward order, so the exchange is a handful of large in-place all-reduces: the buffer is cut into
    ``nchunks`` runs of whole blocks, and the run that ends the network -- the first one the backward pass completes -- is
    sent while the earlier blocks are still being differentiated (RCCL runs on its own stream; xGMI links are
    point-to-point, so few LARGE messages are the right shape).  The ~60 remaining parameters (stem, stage transitions,
    meta-token MLPs, norms, head) travel flattened in one more all-reduce at the end.

        sync = attach_flat_grad_sync(model, opt)     # once; broadcasts rank 0's parameters and buffers
        ...
        opt.zero_grad(); loss.backward(); sync.finish(); opt.step()
    """

    def __init__(self, flat_grad: torch.Tensor, chunk_bounds, rest_params, group=None, force: bool = False, compress: Optional[str] = None, opt=None):
        self.flat = flat_grad
        self._opt = opt                                  # the FlatAdamW that owns `flat_grad` (None in unit tests that drive a bare buffer)
        self.bounds = list(chunk_bounds)                 # [(start, end)] element ranges of flat_grad, in forward order
        self.rest = [p for p in rest_params]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())      # force: run the collectives in a 1-rank group (tests)
        self.compress = compress           # "bf16": the block gradients travel as bfloat16 (half the xGMI bytes)
        self._work = {}
        self._wire = {}                    # chunk -> bf16 wire buffer (compress == "bf16")
        self._tmp = {}                     # chunk -> fp32 staging of gradient / world on its way to the wire
        self._hold = False                 # no_sync(): gradient accumulation in progress, nothing is sent

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context only accumulate locally (like DDP.no_sync()); the
        exchange starts from the first backward pass run OUTSIDE it, whose ``finish()`` sends every chunk.  Without this, a
        chunk would be all-reduced with the partial gradient of the first micro-batch and later micro-batches would add
        local gradients on top of cross-rank sums."""
        prev, self._hold = self._hold, True
        try:
            yield
        finally:
            self._hold = prev

    def chunk_ready(self, k: int) -> None:
        """The backward pass has finished writing chunk k: start its all-reduce (asynchronous w.r.t. the compute stream)."""
        if not self.active or self._hold or k in self._work:
            return
        s, e = self.bounds[k]
        if self.compress == "bf16":
            wire = self._wire.get(k)
            if wire is None:
                wire = self._wire[k] = torch.empty(e - s, device=self.flat.device, dtype=torch.bfloat16)
            # the MEAN is formed on the wire: every rank sends gradient / world (an exact exponent shift for the power-of-two world sizes of one node), so the bf16 partial
            # sums of the ring stay at the magnitude of one rank's gradient whatever the world size is (un-divided bf16 sums lose a bit per doubling of the world)
            torch.mul(self.flat[s:e], 1.0 / self.world, out=self._scratch(k, e - s))
            wire.copy_(self._scratch(k, e - s))
            self._work[k] = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work[k] = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self) -> None:
        """After backward: send whatever has not been sent, wait, average, and exchange the remaining parameters' gradients."""
        if not self.active:
            return
        if self._hold:
            raise RuntimeError("FlatGradSync.finish() inside no_sync(): run the last micro-batch outside the context")
        if self._opt is not None:
            # model.zero_grad() (set_to_none) un-binds p.grad from the flat buffer: the block backward then hands its gradients to autograd,
            # no chunk callback fires, and the flat buffer would be exchanged stale.  Re-bind (copying stray gradients in) BEFORE sending.
            touched = self._opt.rebind_grads()
            # A chunk whose all-reduce already started during the backward pass (some of its parameters were still bound, so its callback fired) must not be
            # rewritten now: its slices would travel un-averaged, or be summed twice if the chunk were sent again -- the ranks would diverge silently.
            for k in self._work:
                s_, e_ = self.bounds[k]
                if any(off < e_ and off + n > s_ for off, n in touched):
                    raise RuntimeError("FlatGradSync.finish(): some block parameters had their .grad cleared or replaced (p.grad = None on a subset) while others of the same "
                                       f"chunk {k} stayed bound to the flat gradient buffer, and that chunk's all-reduce had already started.  Clear gradients with "
                                       "optimizer.zero_grad() / model.zero_grad() for ALL parameters (either way is handled), not for a subset.")
        if self.flat.is_cuda:
            from . import blocks as _blocks
            _blocks.drain_deferred()                     # the weight-gradient side stream has written every slice that is about to travel
        for k in range(len(self.bounds) - 1, -1, -1):
            self.chunk_ready(k)
        grads = [p.grad for p in self.rest if p.grad is not None]
        rest_flat = torch.cat([g.reshape(-1).float() for g in grads]) if grads else None
        rest_work = dist.all_reduce(rest_flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if rest_flat is not None else None
        for k in sorted(self._work):
            self._work[k].wait()
            if self.compress == "bf16":
                s, e = self.bounds[k]
                self.flat[s:e].copy_(self._wire[k])
        self._work.clear()
        if self.compress != "bf16":
            self.flat.mul_(1.0 / self.world)
        if rest_work is not None:
            rest_work.wait()
            rest_flat.mul_(1.0 / self.world)
            off = 0
            with torch.no_grad():
                for g in grads:
                    n = g.numel()
                    g.copy_(rest_flat[off:off + n].view(g.shape))
                    off += n

    def attach_buffer_broadcast(self, model: torch.nn.Module, src: int = 0):
        """broadcast_buffers() in front of every TRAINING forward pass of `model` (a forward pre-hook; returns its handle)."""
        if not self.active:
            return None

        def _pre(mod, args):
            if mod.training and torch.is_grad_enabled():
                self.broadcast_buffers(mod, src)
        self._buffer_hook = model.register_forward_pre_hook(_pre)
        return self._buffer_hook

    def _scratch(self, k: int, n: int) -> torch.Tensor:
        t = self._tmp.get(k)
        if t is None or t.numel() != n:
            t = self._tmp[k] = torch.empty(n, device=self.flat.device, dtype=torch.float32)
        return t

    def broadcast_buffers(self, model: torch.nn.Module, src: int = 0) -> None:
        """The reference's DDP default (main.py:333, broadcast_buffers=True): rank `src`'s buffers -- the BatchNorm running statistics (3 494 elements for LeMeViT-Base) and
        their integer batch counters -- reach every rank before a forward pass, as ONE flat broadcast per element class (DistributedDataParallel._sync_buffers coalesces the same
        way).  attach_flat_grad_sync(broadcast_buffers=True), the default, runs it in front of every training forward pass."""
        if not self.active:
            return
        with torch.no_grad():
            for pick, dt in ((lambda b: b.is_floating_point(), torch.float32), (lambda b: not b.is_floating_point(), torch.int64)):
                bufs = [b for b in model.buffers() if pick(b)]
                if not bufs:
                    continue
                flat = torch.cat([b.reshape(-1).to(dt) for b in bufs])          # (one launch; .to() is a no-op for fp32 / int64 buffers)
                dist.broadcast(flat, src, group=self.group)
                views, off = [], 0
                for b in bufs:
                    n = b.numel()
                    views.append(flat[off:off + n].view(b.shape))
                    off += n
                if all(b.dtype == dt for b in bufs):
                    torch._foreach_copy_(bufs, views)                            # one launch for all of them (12 BatchNorm statistics tensors + 6 counters for LeMeViT)
                else:
                    for b, v in zip(bufs, views):
                        b.copy_(v)


def attach_flat_grad_sync(model: torch.nn.Module, opt, nchunks: int = 4, group=None, src: int = 0, force: bool = False,
                          compress: Optional[str] = None, broadcast_buffers: bool = True) -> FlatGradSync:
    """Wire a FlatAdamW-managed model for data parallelism: broadcast rank `src`'s parameters / buffers, cut the flat gradient
    buffer into `nchunks` runs of whole blocks and hook each run's all-reduce to the backward pass of its first block.
    broadcast_buffers (default True, as the reference's DistributedDataParallel at main.py:333): rank `src`'s buffers are broadcast in front of EVERY training forward pass
    (a forward pre-hook on the model), so the BatchNorm running statistics -- which every rank updates from its own shard of the batch -- stay those of rank `src` everywhere,
    as under torch DDP; False: only once, here."""
    from .optim import flat_chunk_plan
    bounds, first = flat_chunk_plan(model, opt, nchunks)
    block_params = {id(p) for _, p, _, _ in opt._slices}
    rest = [p for p in model.parameters() if p.requires_grad and id(p) not in block_params]
    sync = FlatGradSync(opt._flat_g, bounds, rest, group, force, compress, opt=opt)
    for k, blk in enumerate(first):                                    # chunk k is complete when its FIRST block has been differentiated
        for p in blk.parameters():
            p._lmv_grad_cb = (lambda kk=k: sync.chunk_ready(kk))
    if sync.active:
        with torch.no_grad():
            dist.broadcast(opt._flat_p, src, group=group)
            opt.refresh()
            for p in rest:
                dist.broadcast(p.data, src, group=group)
        sync.broadcast_buffers(model, src)
    if broadcast_buffers:
        sync.attach_buffer_broadcast(model, src)
    return sync
