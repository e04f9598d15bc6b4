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
        from . import ops as _ops
        _ops.check_stage_errors("graph warm-up", sync=False)          # (just synchronised) a lost hand-off in the warm-up runs is an exception, not a captured graph of wrong results
        if pre_capture is not None:
            pre_capture()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            step_fn()
        torch.cuda.synchronize()

    def __call__(self) -> None:
        # the verdict on the stage launches of the replays that have completed so far (a plain load of the pinned error word): a replayed stage kernel that lost a
        # hand-off is an exception on the next call at the latest.  Callers that consume the outputs of the LAST replay run ops.check_stage_errors(sync=True) first.
        from . import ops as _ops
        _ops.check_stage_errors("graph replay", sync=False)
        self.graph.replay()


def split_forward(model: Callable, x: torch.Tensor, parts: int, outs: Optional[list] = None):
    """``model(x)`` as ``parts`` independent sub-batches, each on its own stream forked from (and joined back into) the current one;
    returns the concatenated output (or fills ``outs``).  Inference only: the images of a batch do not interact in eval mode (the
    reference's BatchNorm uses running statistics there, models/lemevit.py:663-676), so this is the same computation -- but the ramp and
    the tail of every kernel of one sub-batch run under another sub-batch's kernels instead of under an idle chip (LeMeViT-Base 224^2,
    B = 128, hipGraph replay: 8.93 -> 8.59 ms with 2 parts, 8.51 with 4; tools/split_infer.py).  Inside ``torch.cuda.graph`` the forks
    become branches of the captured graph.  Scratch of the C-ABI calls is keyed by stream (model._persistent), so the branches share
    nothing but the weights."""
    parts = max(1, min(int(parts), x.shape[0]))
    if parts == 1:
        y = model(x)
        if outs is not None:
            outs[:] = [y]
        return y
    xs = x.chunk(parts)
    cur = torch.cuda.current_stream()
    from .blocks import aux_streams
    streams = aux_streams(x.device, len(xs) - 1)          # shared with the training schedules: main + 3 streams per process (blocks.aux_streams)
    ys = [None] * len(xs)
    # Sub-batch 0 is ISSUED first, on the current stream: every lazily built operand cache of the model (bf16 casts, LayerNorm folds, conv + BatchNorm
    # folds, the classifier tail, the packed S stage) is then filled by kernels on `cur`.  The other streams fork from the point BEFORE it (they
    # run next to it) -- unless it filled a cache (model.cache_fills() moved: the first pass after a weight update): then they wait for the
    # whole of sub-batch 0, so that no stream reads a half-written cached operand.  (Before round 4 the forks came first and the later
    # sub-batches could read caches another stream was still writing.)
    from . import model as _model
    fork = torch.cuda.Event()
    fork.record(cur)
    fills = _model.cache_fills()
    was, _model.launches.concurrent = _model.launches.concurrent, len(xs)      # (persistent stage kernels of a shape share the chip up to a bound: model._sstage_applies; thread-local)
    try:
        ys[0] = model(xs[0])
        cold = _model.cache_fills() != fills
        for i, s in enumerate(streams):
            if cold:
                s.wait_stream(cur)
            else:
                s.wait_event(fork)
            with torch.cuda.stream(s):
                ys[i + 1] = model(xs[i + 1])
                xs[i + 1].record_stream(s)
    finally:
        _model.launches.concurrent = was
    for s, y in zip(streams, ys[1:]):
        cur.wait_stream(s)
        y.record_stream(cur)
    if outs is not None:
        outs[:] = ys
        return None
    return torch.cat(ys)


def try_graphed(step_fn: Callable[[], None], warmup: int = 3, pre_capture: Optional[Callable[[], None]] = None):
    """GraphedStep if capture succeeds, else the eager callable (with the reason)."""
    try:
        return GraphedStep(step_fn, warmup, pre_capture), None
    except Exception as e:  # capture can fail on ops that synchronise; fall back to eager
        torch.cuda.synchronize()
        if "lost an in-launch hand-off" in str(e):
            raise               # wrong tensors in the warm-up runs are not a reason to fall back: they are an error (ops.check_stage_errors)
        return step_fn, f"{type(e).__name__}: {e}"
