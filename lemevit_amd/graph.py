"""hipGraph capture of a whole training (or inference) step.

A LeMeViT-Base train step is ~2 000 short kernel launches; issued eagerly from Python the host cannot keep the
queue full (~10-15 % GPU idle time on an MI355X).  ``GraphedStep`` captures one full step -- zero_grad, autocast
forward through the HIP kernels, loss, hand-written backward, optimizer -- into a hipGraph once and replays it: the
C-ABI kernels are enqueued on ``torch.cuda.current_stream()``, which is the capture stream inside
``torch.cuda.graph``; scratch comes from PyTorch's graph-private pool; DropPath masks and the synthetic targets use
PyTorch's graph-safe Philox generator, so every replay draws fresh randomness.

The reference has no equivalent (eager PyTorch, benchmark.py:572-596); semantics are unchanged -- replaying the graph
is the same kernel sequence as the eager step on the same static input buffers.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GraphedStep:
    """Capture ``step_fn()`` (a closure over static tensors) after ``warmup`` eager runs; ``__call__`` replays it.

    Requirements on ``step_fn``: static shapes, no host synchronisation (``.item()``, prints of tensors), optimizers
    constructed with ``capturable=True``.  Outputs the caller wants to read must be written to tensors the closure
    owns (they are overwritten on every replay)."""

    def __init__(self, step_fn: Callable[[], None], warmup: int = 3, pre_capture: Optional[Callable[[], None]] = None):
        self._fn = step_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if pre_capture is not None:
            pre_capture()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            step_fn()
        torch.cuda.synchronize()

    def __call__(self) -> None:
        self.graph.replay()


def try_graphed(step_fn: Callable[[], None], warmup: int = 3, pre_capture: Optional[Callable[[], None]] = None):
    """GraphedStep if capture succeeds, else the eager callable (with the reason)."""
    try:
        return GraphedStep(step_fn, warmup, pre_capture), None
    except Exception as e:  # capture can fail on ops that synchronise; fall back to eager
        torch.cuda.synchronize()
        return step_fn, f"{type(e).__name__}: {e}"
