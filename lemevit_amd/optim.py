"""Flat multi-tensor AdamW for the LeMeBlock parameters (benchmark.py:559-561,587).

``torch.optim.AdamW(fused=True)`` walks the ~450 block parameters of LeMeViT-Base in 16 multi-tensor launches, and every
training pass then re-casts the 146 weight matrices to bf16 for the kernels (fused optimizers update in place without
bumping ``Tensor._version``, so those casts cannot be cached: see model.compute_copy).  Here the block parameters live in
ONE flat fp32 buffer, their gradients in ONE flat fp32 buffer that the block backward writes straight into
(``_BlockFn`` accumulates into ``p.grad`` and hands autograd ``None``), and one launch of ``lmv_adamw_flat`` updates
everything and refreshes the bf16 operand copies in the same pass.  The handful of non-block parameters (stem, stage
transitions, meta-token MLPs, norms, head) stay on ``torch.optim.AdamW(fused=True)``.

Not for DistributedDataParallel: DDP's reducer waits for autograd's AccumulateGrad hooks, which never fire for
gradients written in place -- wrap the model in DDP with a regular optimizer instead (bench.py does that for N > 1).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

Tensor = torch.Tensor
_ALIGN = 8          # elements: 32-byte fp32 / 16-byte bf16 alignment of every parameter's slice


def flat_chunk_plan(model: nn.Module, opt: "FlatAdamW", nchunks: int):
    """Cut the flat buffers of `opt` into `nchunks` runs of whole blocks (forward order = layout order).  Returns (bounds, first): bounds[k] = (start, end) element range,
    first[k] = the LeMeBlock whose backward pass completes chunk k (its first block: the backward pass runs the blocks last to first).  Chunk sizes grow 1 : 2 : ... : nchunks
    in forward order: the backward pass completes the chunks last-to-first, so the big ones (late stages hold most of the parameters anyway) are consumed -- exchanged by
    FlatGradSync, applied by the overlapped FlatAdamW -- under the rest of the backward pass, and the one that cannot overlap anything (the first blocks, differentiated last)
    is the smallest."""
    from .model import LeMeBlock
    blocks = [(name, mod) for name, mod in model.named_modules() if isinstance(mod, LeMeBlock)]
    starts = {}
    for pname, p, off, n in opt._slices:
        for bname, _ in blocks:
            if pname.startswith(bname + "."):
                starts.setdefault(bname, off)
                break
    order = [b for b, _ in blocks if b in starts]
    total = opt._flat_g.numel()
    offs = [starts[b] for b in order] + [total]
    nchunks = max(1, min(nchunks, len(order)))
    tri = nchunks * (nchunks + 1) / 2
    targets = [total * (j * (j + 1) / 2) / tri for j in range(1, nchunks)]
    cuts = [0]
    for i in range(1, len(order)):
        if len(cuts) < nchunks and offs[i] >= targets[len(cuts) - 1]:
            cuts.append(i)
    bounds = [(offs[c], offs[cuts[j + 1]] if j + 1 < len(cuts) else total) for j, c in enumerate(cuts)]
    mods = dict(blocks)
    return bounds, [mods[order[c]] for c in cuts]


def _default_no_decay(name: str, p: Tensor) -> bool:
    return p.ndim <= 1          # biases, LayerNorm / BatchNorm affine, as benchmark.py's create_optimizer_v2 (filter_bias_and_bn)


class FlatAdamW:
    """AdamW over ``model``: block parameters flat + fused into one launch, everything else ``torch.optim.AdamW(fused=True)``.

    Interface: ``step()``, ``zero_grad()``, ``param_groups`` (one dict per group with ``lr`` -- schedulers may edit it),
    ``state_dict()`` / ``load_state_dict()``, ``refresh()`` (re-derive the bf16 copies after the parameters were
    changed by something else; done automatically after ``model.load_state_dict``)."""

    def __init__(self, model: nn.Module, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, no_decay: Callable[[str, Tensor], bool] = _default_no_decay, capturable: bool = True):
        from .model import LeMeBlock, _is_matrix
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        flat_named: List[Tuple[str, nn.Parameter, bool]] = []        # (name, param, wants bf16 copy)
        seen = set()
        for mname, mod in model.named_modules():
            if isinstance(mod, LeMeBlock):
                for pname, p in mod.named_parameters():
                    if p.requires_grad and id(p) not in seen and p.dtype == torch.float32 and p.is_cuda:
                        seen.add(id(p))
                        flat_named.append((f"{mname}.{pname}", p, _is_matrix(pname)))
        if not flat_named:
            raise ValueError("FlatAdamW: the model has no fp32 LeMeBlock parameters on the GPU")
        dev = flat_named[0][1].device
        offs, total = [], 0
        for _, p, _m in flat_named:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self._flat_p = torch.zeros(total, device=dev)
        self._flat_g = torch.zeros(total, device=dev)
        self._exp_avg = torch.zeros(total, device=dev)
        self._exp_avg_sq = torch.zeros(total, device=dev)
        self._wd_mask = torch.zeros(total, device=dev)
        self._shadow = torch.zeros(total, device=dev, dtype=torch.bfloat16)
        self._step_dev = torch.zeros((), device=dev, dtype=torch.int32)
        self._slices: List[Tuple[str, nn.Parameter, int, int]] = []
        self._grad_views: List[Tensor] = []
        with torch.no_grad():
            for (name, p, matrix), off in zip(flat_named, offs):
                n = p.numel()
                self._flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self._flat_p[off:off + n].view(p.shape)
                p.grad = self._flat_g[off:off + n].view(p.shape)
                self._grad_views.append(p.grad)
                p._lmv_flat_grad = True                               # _BlockFn.backward accumulates into p.grad in place
                if not no_decay(name, p):
                    self._wd_mask[off:off + n] = 1.0
                if matrix:
                    p._lmv_shadow = self._shadow[off:off + n].view(p.shape)
                self._slices.append((name, p, off, n))
        # transposed bf16 copies of the mlp.3 weights of the blocks whose fc2 dX runs on the register-stationary GEMM (C = 192 / 384 and a
        # hidden width that is a multiple of 64, csrc/rsgemm.hip): refreshed by ONE lmv_transpose_batch launch behind every update
        self._tpairs: List[Tuple[Tensor, Tensor]] = []
        # ... and of mlp.0 / attn.qkv / attn.proj of the C = 384 "S" blocks, whose dX then runs on the whole-width kernel (csrc/wngemm.hip)
        for name, p, off, n in self._slices:
            fc2 = name.endswith("mlp.3.weight") and p.dim() == 2 and p.shape[0] in (192, 384) and p.shape[1] % 64 == 0 and p.shape[1] >= 512
            wide = p.dim() == 2 and p.shape[1] == 384 and p.shape[0] % 64 == 0 and name.endswith(("mlp.0.weight", "attn.qkv.weight", "attn.proj.weight"))
            if fc2 or wide:
                wt = torch.empty((p.shape[1], p.shape[0]), device=dev, dtype=torch.bfloat16)
                p._lmv_shadow_t = wt
                self._tpairs.append((p._lmv_shadow, wt))
        self.refresh()
        rest_decay = [p for n, p in model.named_parameters() if p.requires_grad and id(p) not in seen and not no_decay(n, p)]
        rest_plain = [p for n, p in model.named_parameters() if p.requires_grad and id(p) not in seen and no_decay(n, p)]
        groups = [g for g in (dict(params=rest_decay, weight_decay=weight_decay), dict(params=rest_plain, weight_decay=0.0)) if g["params"]]
        self._rest = torch.optim.AdamW(groups, lr=lr, betas=betas, eps=eps, fused=True, capturable=capturable) if groups else None
        # 'params' lists the flat-managed parameters so that code walking param_groups (GradScaler.unscale_, timm's clipping,
        # schedulers) sees every parameter; their .grad tensors are views of the flat gradient buffer
        self.param_groups = [dict(params=[p for _, p, _, _ in self._slices], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                  name="lemevit_blocks_flat")] + (self._rest.param_groups if self._rest else [])
        self._hook = model.register_load_state_dict_post_hook(lambda *_: self.refresh())

    # ---- the optimizer interface ---------------------------------------------------------------------------------
    def _rebind(self, keep: bool) -> list:
        """Every flat-managed ``p.grad`` must be a view of the flat gradient buffer.  ``model.zero_grad()`` (set_to_none=True) or
        ``p.grad = None`` breaks that: the block backward then hands the gradients to autograd, which allocates fresh ``.grad``
        tensors the fused update would never read.  keep=True copies such a stray gradient into its slice first (COPY, not add: a ``.grad`` that
        is not our view means the caller cleared or replaced the gradients since the slice was last bound, so whatever the slice still holds
        is stale -- e.g. the previous step's gradients after ``model.zero_grad()`` -- and autograd has accumulated every micro-batch since then
        into the stray tensor)."""
        touched = []                                                   # (offset, length) of every slice whose CONTENT this call changed
        for i, (_, p, off, n) in enumerate(self._slices):
            g = p.grad
            if g is self._grad_views[i]:                               # the common case: one identity test per parameter
                continue
            view = self._grad_views[i]
            if g is not None and g.data_ptr() != view.data_ptr():
                if keep:
                    view.copy_(g.detach().to(view.dtype).view(view.shape))
                    touched.append((off, n))
            elif g is None and keep:
                view.zero_()                                               # cleared and not re-computed: no gradient (the slice would be stale)
                touched.append((off, n))
            p.grad = view
        return touched

    def zero_grad(self, set_to_none: bool = True) -> None:
        self._flat_g.zero_()                                           # the flat gradients stay allocated: the kernels accumulate into them
        self._rebind(keep=False)
        if self._rest is not None:
            self._rest.zero_grad(set_to_none=set_to_none)

    def _apply(self, s: int, e: int) -> None:
        g0 = self.param_groups[0]          # schedulers / users may edit any of these (as for torch.optim.AdamW)
        b1, b2 = g0["betas"]
        wd = self._wd_mask[s:e]
        ops.adamw_flat(self._flat_p[s:e], self._flat_g[s:e], self._exp_avg[s:e], self._exp_avg_sq[s:e], wd, float(g0["lr"]),
                       float(b1), float(b2), float(g0["eps"]), float(g0["weight_decay"]), 0, shadow=self._shadow[s:e], step_dev=self._step_dev)

    def rebind_grads(self) -> list:
        """Public form of ``_rebind(keep=True)``: call before anything reads the flat gradient buffer directly (FlatGradSync.finish does).  Returns the (offset, length)
        slices of the flat buffer it rewrote."""
        return self._rebind(keep=True)

    @torch.no_grad()
    def step(self) -> None:
        from . import blocks as _blocks
        _blocks.drain_deferred()           # backstop: the weight-gradient side stream must have been joined before the update reads the gradients
        self._rebind(keep=True)
        if self._rest is not None:
            self._rest.step()
        self._step_dev += 1
        self._apply(0, self._flat_p.numel())
        ops.transpose_batch(self._tpairs)

    @torch.no_grad()
    def refresh(self) -> None:
        """bf16 operand copies (and their transposed forms) <- current fp32 parameters."""
        self._shadow.copy_(self._flat_p)
        ops.transpose_batch(self._tpairs)

    def state_dict(self) -> Dict[str, object]:
        return dict(step=int(self._step_dev.item()), exp_avg=self._exp_avg.clone(), exp_avg_sq=self._exp_avg_sq.clone(),
                    names=[(n, off, k) for n, _, off, k in self._slices], lr=self.param_groups[0]["lr"], betas=tuple(self.param_groups[0]["betas"]),
                    eps=self.param_groups[0]["eps"], weight_decay=self.param_groups[0]["weight_decay"],
                    rest=None if self._rest is None else self._rest.state_dict())

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        if [(n, off, k) for n, _, off, k in self._slices] != list(sd["names"]):
            raise ValueError("FlatAdamW.load_state_dict: parameter layout differs")
        self._step_dev.fill_(int(sd["step"]))
        self._exp_avg.copy_(sd["exp_avg"]); self._exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0]["lr"] = sd["lr"]
        for k in ("betas", "eps", "weight_decay"):
            if k in sd:
                self.param_groups[0][k] = sd[k]
        if self._rest is not None and sd.get("rest") is not None:
            self._rest.load_state_dict(sd["rest"])


class ModelEma:
    """Exponential moving average of a model's weights -- the reference's optional ``--model-ema`` (``timm.utils.ModelEmaV2``: main.py:316 builds it, engine.py calls
    ``model_ema.update(model)`` after every optimizer step, validate runs on ``model_ema.module``).

    ``module`` is an eval-mode copy of the model.  With a ``FlatAdamW`` the block parameters of the copy are views of ONE flat fp32 buffer laid out like the optimizer's, and
    their update is a single ``lmv_ema_flat`` launch; the ~60 remaining parameters and the buffers (BatchNorm running statistics) go through one multi-tensor lerp, integer
    buffers are copied -- the same update rule as ModelEmaV2 (every ``state_dict`` entry: ``ema = decay * ema + (1 - decay) * model``)."""

    def __init__(self, model: nn.Module, decay: float = 0.9998, opt: Optional[FlatAdamW] = None):
        import copy
        self.decay = float(decay)
        self.module = copy.deepcopy(model)
        self.module.eval()
        for p in self.module.parameters():
            p.requires_grad_(False)
            p.grad = None
            for attr in ("_lmv_shadow", "_lmv_shadow_t", "_lmv_flat_grad", "_lmv_grad_cb"):          # the copy is a plain model: no optimizer-owned operand copies, no gradient hooks
                if hasattr(p, attr):
                    delattr(p, attr)
        self._flat_src: Optional[Tensor] = None
        self._flat_ema: Optional[Tensor] = None
        flat_names = set()
        if opt is not None:
            self._flat_src = opt._flat_p
            self._flat_ema = opt._flat_p.detach().clone()
            table = dict(self.module.named_parameters())
            for name, _, off, n in opt._slices:
                q = table[name]
                q.data = self._flat_ema[off:off + n].view(q.shape)
                flat_names.add(name)
        src = model.state_dict()
        self._pairs_f: List[Tuple[Tensor, str]] = []
        self._pairs_i: List[Tuple[Tensor, str]] = []
        for k, v in self.module.state_dict().items():
            if k in flat_names:
                continue
            if k not in src:
                raise KeyError(k)
            (self._pairs_f if v.dtype.is_floating_point else self._pairs_i).append((v, k))
        self._src_cache: Optional[Tuple[int, List[Tensor], List[Tensor]]] = None

    def _sources(self, model: nn.Module) -> Tuple[List[Tensor], List[Tensor]]:
        """The source tensors of the non-flat entries, looked up ONCE per source module (parameters and buffers are updated in place, so the tensors stay the same objects):
        the reference's flow builds the EMA from the bare model and then calls update() with the DistributedDataParallel wrapper (main.py:316, engine.py) -- the wrapper's
        state_dict keys carry a 'module.' prefix, so the wrapper is peeled first (timm's ModelEmaV2 zips the two state_dicts positionally for the same reason)."""
        bare = model
        while hasattr(bare, "module") and isinstance(getattr(bare, "module"), nn.Module) and not isinstance(bare, type(self.module)):
            bare = bare.module
        ent = self._src_cache
        if ent is None or ent[0] != id(bare):
            src = bare.state_dict()
            missing = [k for _, k in self._pairs_f + self._pairs_i if k not in src]
            if missing:
                raise KeyError(f"ModelEma.update: the model has no state_dict entry {missing[0]!r} (and {len(missing) - 1} more)")
            ent = self._src_cache = (id(bare), [src[k] for _, k in self._pairs_f], [src[k] for _, k in self._pairs_i])
        return ent[1], ent[2]

    @torch.no_grad()
    def update(self, model: nn.Module) -> None:
        from . import model as _model
        if self._flat_ema is not None:
            ops.ema_flat(self._flat_ema, self._flat_src, self.decay)
        src_f, src_i = self._sources(model)
        if self._pairs_f:
            torch._foreach_lerp_([e for e, _ in self._pairs_f], [t.detach() if t.dtype == e.dtype else t.detach().to(e.dtype) for (e, _), t in zip(self._pairs_f, src_f)], 1.0 - self.decay)
        for (e, _), t in zip(self._pairs_i, src_i):
            e.copy_(t)
        _model.new_training_pass()          # the native launch writes through raw pointers (no Tensor._version bump): drop every cached operand copy of the EMA module
