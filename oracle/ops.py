"""oracle/ops.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end of oracle/pvcnn_ops_ref.c exposing the Python-visible surface of the
reference plugin `_pvcnn_backend` (reference: experiments/model/pvcnn/modules/functional/
src/bindings.cpp:10-37) on CPU torch tensors.  Used by tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg and oracle/gen_golden.py (where it is the stand-in for the
CUDA-only plugin when the reference's own nn.Modules are imported).
Nothing under bdm_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_ops.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pvcnn_ops_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_ops.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fp(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def _ip(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def furthest_point_sampling(coords, num_samples):
    b, _, n = coords.shape
    idx = torch.zeros(b, num_samples, dtype=torch.int32)
    lib().oracle_furthest_point_sampling(b, n, int(num_samples), _fp(coords), _ip(idx))
    return idx


def gather_features_forward(features, indices):
    b, c, n = features.shape
    m = indices.shape[1]
    out = torch.zeros(b, c, m)
    lib().oracle_gather_features(b, c, n, m, _fp(features), _ip(indices), _fp(out))
    return out


def ball_query(centers, points, radius, num_neighbors):
    b, _, m = centers.shape
    n = points.shape[2]
    out = torch.zeros(b, m, num_neighbors, dtype=torch.int32)
    lib().oracle_ball_query(b, n, m, ctypes.c_float(radius), int(num_neighbors), _fp(centers),
                            _fp(points), _ip(out))
    return out


def grouping_forward(features, indices):
    b, c, n = features.shape
    _, m, u = indices.shape
    out = torch.zeros(b, c, m, u)
    lib().oracle_grouping(b, c, n, m, u, _fp(features), _ip(indices), _fp(out))
    return out


def three_nearest_neighbors_interpolate_forward(points, centers, features):
    b, c, m = features.shape
    n = points.shape[2]
    idx = torch.zeros(b, 3, n, dtype=torch.int32)
    w = torch.zeros(b, 3, n)
    out = torch.zeros(b, c, n)
    lib().oracle_three_nn_interpolate(b, c, m, n, _fp(points), _fp(centers), _fp(features),
                                      _ip(idx), _fp(w), _fp(out))
    return [out, idx, w]


def avg_voxelize_forward(features, coords, resolution):
    b, c, n = features.shape
    r = int(resolution)
    out = torch.zeros(b, c, r ** 3)
    ind = torch.zeros(b, n, dtype=torch.int32)
    cnt = torch.zeros(b, r ** 3, dtype=torch.int32)
    lib().oracle_avg_voxelize(b, c, n, r, _fp(features), _ip(coords), _fp(out), _ip(ind), _ip(cnt))
    return [out, ind, cnt]


def trilinear_devoxelize_forward(resolution, is_training, coords, features):
    b, c, _ = features.shape
    n = coords.shape[2]
    out = torch.zeros(b, c, n)
    if is_training:
        inds, wgts = torch.zeros(b, 8, n, dtype=torch.int32), torch.zeros(b, 8, n)
        lib().oracle_trilinear_devoxelize_training(b, c, n, int(resolution), _fp(coords), _fp(features), _fp(out), _ip(inds), _fp(wgts))
        return [out, inds, wgts]
    lib().oracle_trilinear_devoxelize(b, c, n, int(resolution), _fp(coords), _fp(features), _fp(out))
    # eval mode returns 1-element placeholders (trilinear_devox.cpp:45-53)
    return [out, torch.zeros(1, dtype=torch.int32), torch.zeros(1)]


def gather_features_backward(grad_y, indices, n):
    b, c, m = grad_y.shape
    gx = torch.zeros(b, c, int(n))
    lib().oracle_gather_features_backward(b, c, int(n), m, _fp(grad_y), _ip(indices), _fp(gx))
    return gx


def grouping_backward(grad_y, indices, n):
    b, c, m, u = grad_y.shape
    gx = torch.zeros(b, c, int(n))
    lib().oracle_grouping_backward(b, c, int(n), m, u, _fp(grad_y), _ip(indices), _fp(gx))
    return gx


def three_nearest_neighbors_interpolate_backward(grad_y, indices, weights, m):
    b, c, n = grad_y.shape
    gx = torch.zeros(b, c, int(m))
    lib().oracle_three_nn_interpolate_backward(b, c, n, int(m), _fp(grad_y), _ip(indices), _fp(weights), _fp(gx))
    return gx


def trilinear_devoxelize_backward(grad_y, indices, weights, resolution):
    b, c, n = grad_y.shape
    r3 = int(resolution) ** 3
    gx = torch.zeros(b, c, r3)
    lib().oracle_trilinear_devoxelize_backward(b, c, n, r3, _ip(indices), _fp(weights), _fp(grad_y), _fp(gx))
    return gx


def avg_voxelize_backward(grad_y, indices, cnt):
    b, c, s = grad_y.shape
    n = indices.shape[1]
    gx = torch.zeros(b, c, n)
    lib().oracle_avg_voxelize_backward(b, c, n, s, _ip(indices), _ip(cnt), _fp(grad_y), _fp(gx))
    return gx
