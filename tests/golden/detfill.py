"""Bit-portable deterministic tensors for golden fixtures.

Weights and inputs of every golden case are *generated*, not stored: an integer hash
(splitmix64 over the flat element index, keyed by a CRC of the tensor's name) mapped to 24-bit
uniforms in [-1, 1).  Integer arithmetic only, so the numbers are identical in the build
container (where ``gen_golden.py`` runs the reference) and on the GPU box (where the tests
rebuild them).  Fixtures therefore hold only expected outputs.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Tuple

import numpy as np
import torch


def det_uniform(n: int, seed: int) -> np.ndarray:
    """n float64 values in [-1, 1), each exactly representable in float32."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return u * 2.0 - 1.0


def det_tensor(shape, name: str, seed: int = 0, scale: float = 1.0, offset: float = 0.0,
               dtype=torch.float32) -> torch.Tensor:
    n = int(math.prod(shape)) if len(shape) else 1
    key = (zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF
    a = det_uniform(n, key) * scale + offset
    return torch.from_numpy(a).to(dtype).reshape(tuple(shape))


def fill_state_dict(spec: Dict[str, Tuple[int, ...]], seed: int, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic, *non-degenerate* values for every entry of a LeMeViT state_dict:
    Linear/conv weights uniform with std 1/sqrt(fan_in) (so softmax logits are O(1) and scale
    bugs show), LN/BN affines away from (1, 0), BN running stats away from (0, 1)."""
    sd: Dict[str, torch.Tensor] = {}
    for k, shp in spec.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_mean"):
            sd[k] = det_tensor(shp, k, seed, 0.2, 0.0, dtype)
        elif k.endswith("running_var"):
            sd[k] = det_tensor(shp, k, seed, 0.5, 1.0, dtype)
        elif k == "meta_tokens":
            sd[k] = det_tensor(shp, k, seed, 1.0, 0.0, dtype)
        elif len(shp) >= 2:
            fan_in = int(math.prod(shp[1:]))
            sd[k] = det_tensor(shp, k, seed, math.sqrt(3.0 / fan_in), 0.0, dtype)
        elif k.endswith(".weight"):      # LN / BN gamma
            sd[k] = det_tensor(shp, k, seed, 0.2, 1.0, dtype)
        else:                            # biases, LN / BN beta
            sd[k] = det_tensor(shp, k, seed, 0.1, 0.0, dtype)
    return sd


def sample_idx(numel: int, n: int = 4096) -> np.ndarray:
    """Deterministic sample positions into a flattened tensor (all of it when small)."""
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    return np.linspace(0, numel - 1, n).astype(np.int64)


def sample(t: torch.Tensor, n: int = 4096) -> np.ndarray:
    flat = t.detach().reshape(-1).to(torch.float64).cpu().numpy()
    return flat[sample_idx(flat.size, n)].astype(np.float32)
