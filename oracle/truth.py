"""oracle/truth.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The denoisers of oracle/ref_net.py evaluated in float64 on fp32-determined geometry: an error BUDGET tool, not a parity oracle.

Two fp32 implementations of one network (the torch-CPU oracle and the HIP path) each sit ~1e-6 away from the exact real-arithmetic
result, so their mutual distance says nothing about which one carries how much of it.  Here every FEATURE tensor and every weight is
float64, while every discrete decision and every geometric quantity stays exactly what the fp32 reference computes: furthest point
sampling, ball-query indices, 3-NN indices AND weights, voxel coordinates, voxel assignment and counts, trilinear corner indices and
weights all come from the C restatement (oracle/pvcnn_ops_ref.c) on the fp32 coordinates.  The result is "the reference's function
evaluated without rounding in the feature arithmetic"; `rel_l2(hip, truth)` and `rel_l2(oracle32, truth)` then split the HIP-vs-oracle
distance into its two halves (tools/error_budget.py).

Only tests/ and tools/ import this.
"""
import contextlib

import torch
import torch.nn.functional as F

from . import ops as O
from . import ref_net as RN


def _grouping64(features, idx):
    B, C, _ = features.shape
    M, U = idx.shape[1:]
    flat = idx.reshape(B, 1, M * U).long().expand(-1, C, -1)
    return torch.gather(features, 2, flat).reshape(B, C, M, U)


def _three_nn64(points, centers, features):
    _, idx, w = _REAL["three_nearest_neighbors_interpolate_forward"](points.float().contiguous(), centers.float().contiguous(),
                                                                     features.float().contiguous())
    B, C, _ = features.shape
    n = idx.shape[2]
    out = torch.zeros(B, C, n, dtype=features.dtype)
    for k in range(3):   # out = f[i0] w0 + f[i1] w1 + f[i2] w2 (neighbor_interpolate.cu:96-116), exact products of the fp32 weights
        out = out + torch.gather(features, 2, idx[:, k].long()[:, None].expand(-1, C, -1)) * w[:, k].to(features.dtype)[:, None]
    return [out, idx, w]


def _avg_voxelize64(features, coords, r):
    _, ind, cnt = _REAL["avg_voxelize_forward"](features.float().contiguous(), coords, r)
    B, C, _ = features.shape
    out = torch.zeros(B, C, r ** 3, dtype=features.dtype)
    for b in range(B):
        i = ind[b].long()
        out[b].index_add_(1, i, features[b] / cnt[b, i].to(features.dtype)[None])
    return [out, ind, cnt]


def _devox64(r, is_training, coords, features):
    _, inds, wgts = _REAL["trilinear_devoxelize_forward"](r, True, coords.float().contiguous(), features.float().contiguous())
    B, C, _ = features.shape
    out = torch.zeros(B, C, coords.shape[2], dtype=features.dtype)
    for k in range(8):
        out = out + torch.gather(features, 2, inds[:, k].long()[:, None].expand(-1, C, -1)) * wgts[:, k].to(features.dtype)[:, None]
    return [out, inds, wgts]


_REAL = {}


@contextlib.contextmanager
def float64_features():
    """Inside: oracle.ref_net's forwards accept float64 features / weights; geometry stays on the fp32 C restatement."""
    names = ["grouping_forward", "three_nearest_neighbors_interpolate_forward", "avg_voxelize_forward", "trilinear_devoxelize_forward",
             "furthest_point_sampling", "gather_features_forward", "ball_query"]
    for nm in names:
        _REAL[nm] = getattr(O, nm)
    real_vc, real_embed, real_ts = RN.voxel_coords, RN.embedf, RN.timestep_embedding

    def grouping(features, idx):
        if features.dtype == torch.float64:
            return _grouping64(features, idx)
        return _REAL["grouping_forward"](features, idx)

    O.grouping_forward = grouping
    O.three_nearest_neighbors_interpolate_forward = _three_nn64
    O.avg_voxelize_forward = _avg_voxelize64
    O.trilinear_devoxelize_forward = _devox64
    O.furthest_point_sampling = lambda c, m: _REAL["furthest_point_sampling"](c.float().contiguous(), m)
    O.gather_features_forward = lambda f, i: _REAL["gather_features_forward"](f.float().contiguous(), i).to(f.dtype)
    O.ball_query = lambda c, p, radius, u: _REAL["ball_query"](c.float().contiguous(), p.float().contiguous(), radius, u)
    # voxel coordinates are geometry: computed in fp32 exactly as the reference does (voxelization.py:16-25)
    RN.voxel_coords = lambda coords, r: real_vc(coords.float(), r)
    RN.embedf = lambda sd, pre, t, dim: real_embed(sd, pre, t, dim)
    RN.timestep_embedding = lambda t, dim: real_ts(t, dim).double()
    try:
        yield
    finally:
        for nm in names:
            setattr(O, nm, _REAL[nm])
        RN.voxel_coords, RN.embedf, RN.timestep_embedding = real_vc, real_embed, real_ts


def pvcnn_forward_f64(sd, inputs, t, prefix="", **kw):
    """ref_net.pvcnn_forward with float64 features and weights (see the module docstring) -> float64 (B, 3, N)."""
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with float64_features():
        return RN.pvcnn_forward(sd64, inputs.double(), t, prefix=prefix, **kw)
