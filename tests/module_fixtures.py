"""Seeded inputs shared by oracle/gen_golden_modules.py (which runs the REFERENCE's modules on them) and the tests that
compare the oracle (CPU) and the HIP modules (GPU) with tests/golden/modules.npz."""
import os

import numpy as np
import torch

from helpers import seeded

B, N, M, U, TE, R, RADIUS, WEIGHT_SEED = 2, 300, 64, 16, 8, 8, 0.3, 17
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modules.npz")


def make_inputs():
    x = {}
    x["coords"] = (seeded((B, 3, N), 501) * 0.25).contiguous()
    x["feat16"] = seeded((B, 16, N), 502)
    x["feat32"] = seeded((B, 32, N), 503)
    x["cfeat"] = seeded((B, 32, M), 504)                     # features living on the M centres
    te = seeded((B, TE), 505)
    x["temb"] = te[:, :, None].expand(-1, -1, N).contiguous()  # point-invariant, as the denoisers' t_emb (pvcnn.py:88)
    x["ctemb"] = te[:, :, None].expand(-1, -1, M).contiguous()
    x["grid"] = seeded((B, 16, 4, 4, 4), 506)
    x["grouped16"] = seeded((B, 16, 10, 6), 507)
    x["grid_r"] = seeded((B, 16, R ** 3), 508)                # a (B, C, R^3) voxel grid for the devoxelisation wrapper
    return x


def load():
    return np.load(GOLD, allow_pickle=False)


def state_dict(g, name):
    """Procedural weights of fixture `name`, keyed as the reference module's state dict."""
    from bdm_amd.utils.procedural import procedural_tensor
    keys = [str(k) for k in g[name + "__keys"]]
    shapes = [eval(s) for s in g[name + "__shapes"]]
    return {k: procedural_tensor(name + "." + k, sh, int(g["weight_seed"])) for k, sh in zip(keys, shapes)}


def outs(g, name):
    res, i = [], 0
    while f"{name}__out{i}" in g:
        res.append(torch.from_numpy(g[f"{name}__out{i}"]))
        i += 1
    return res
