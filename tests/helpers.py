import numpy as np
import pytest
import torch

# The superseded kernel families (fp32-MFMA convolution / flash attention, one-kernel sparse convolution) live in
# bdm_amd/csrc/experimental and are built by `make EXPERIMENTAL=1` only; tests of them run against such a build:
#   make -C bdm_amd/csrc EXPERIMENTAL=1 OUT=../libbdm_hip_experimental.so EXT=/tmp/_pvcnn_backend_exp.so
#   BDM_LIB_PATH=bdm_amd/libbdm_hip_experimental.so python -m pytest tests -m gpu
experimental = pytest.mark.skipif("not __import__('bdm_amd._lib', fromlist=['x']).has_experimental()",
                                  reason="kernel family of the EXPERIMENTAL=1 build")


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


def golden_trajectory(name):
    """tests/golden/traj_<name>.npz (oracle/gen_golden_traj.py): case description + the oracle's final cloud."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"traj_{name}.npz"))
    return {k: g[k] for k in g.files}
