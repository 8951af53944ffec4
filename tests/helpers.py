import numpy as np
import torch


def seeded(shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))


def point_cloud_inputs(B, C, N, seed):
    """Same draw as oracle/gen_golden.py: xyz ~ 0.5*N(0,1), features ~ N(0,1)."""
    x = seeded((B, C, N), seed)
    x[:, :3] *= 0.5
    return x


def rel_l2(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
