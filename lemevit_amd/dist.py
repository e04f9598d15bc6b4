"""Data parallelism for the LeMeViT hot path: one process per GPU, images sharded over ranks, ONE exchange per step --
the gradient all-reduce (reference: main.py:333 NativeDDP, launched by scripts/train.sh:3-8).

On ROCm the "nccl" backend IS RCCL; on an MI355X node the transport is xGMI (fully connected, 7 links x ~153 GB/s per
GPU).  LeMeViT-Base has 53.1 M parameters = 212 MB of fp32 gradients: compressing the buckets to bf16 halves the bytes
on the per-link-bound ring, and a few LARGE buckets (default 100 MB here, vs torch's 25 MB) suit point-to-point xGMI better
than many small ones.  Everything runs on torch.distributed so the same code is testable with gloo on CPU.
"""
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
