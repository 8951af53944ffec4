"""Procedural ("random-init") weights keyed by state-dict key name.

BASELINE.json's configs run on random-init weights of the reference architecture (no
checkpoints are reachable offline).  A tensor's values depend only on (seed, key, shape),
so the reference's modules, the CPU oracle and the HIP path can be given identical
weights without shipping a 112 MB state dict, independent of module construction order
(SURVEY.md 8c).
"""
import zlib

import numpy as np
import torch


def procedural_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    rng = np.random.Generator(np.random.PCG64([zlib.crc32(key.encode()), seed]))
    if len(shape) >= 2:  # conv / linear weight: unit-variance-preserving
        fan_in = int(np.prod(shape[1:]))
        w = rng.standard_normal(shape, dtype=np.float32) / np.float32(np.sqrt(max(fan_in, 1)))
    elif key.endswith("weight"):  # GroupNorm scale
        w = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
    else:  # any bias
        w = 0.05 * rng.standard_normal(shape, dtype=np.float32)
    return torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))


def procedural_state_dict(shapes: dict, seed: int = 0) -> dict:
    """shapes: {key: shape}  ->  {key: tensor}."""
    return {k: procedural_tensor(k, s, seed) for k, s in shapes.items()}


def fill_module_(module: torch.nn.Module, seed: int = 0, prefix: str = "") -> torch.nn.Module:
    """Overwrite every parameter/buffer of `module` in place with its procedural value."""
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if v.is_floating_point():
                v.copy_(procedural_tensor(prefix + k, v.shape, seed).to(v.device))
    return module
