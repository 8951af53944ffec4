"""`_backend`: the reference plugin's Python surface on top of libbdm_hip.so.

Same 12 names, argument order and return conventions as the pybind11 module
`_pvcnn_backend` (reference: experiments/model/pvcnn/modules/functional/src/bindings.cpp:10-37),
so the reference's functional/*.py wrappers work unchanged against this object, including the five
gradient operators and the training-mode devoxelisation (csrc/backward_ops.hip; training is outside the
sampling hot path, they complete the plugin surface).
Outputs are freshly allocated on the inputs' device (callee allocates, caller owns), as in
the reference's at::Tensor API; argument errors raise RuntimeError, nothing exits the process.
"""
import torch

from .. import _lib as L


def _chk_f(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # utils.hpp:7
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")  # utils.hpp:10
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")  # utils.hpp:16


def _chk_i(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")  # utils.hpp:13


class _Backend:
    # sampling.cpp:43-58
    @staticmethod
    def furthest_point_sampling(coords, num_samples):
        _chk_f(coords, "coords")
        b, _, n = coords.shape
        idx = torch.empty(b, int(num_samples), dtype=torch.int32, device=coords.device)   # (every entry is written by the kernel)
        if int(num_samples) <= 0:
            return idx
        L.check(L.lib().bdm_furthest_point_sampling(b, n, int(num_samples), L.ptr(coords), L.ptr(idx),
                                                    L.ptr(None), L.stream()), "furthest_point_sampling")
        return idx

    # sampling.cpp:6-23
    @staticmethod
    def gather_features_forward(features, indices):
        _chk_f(features, "features"); _chk_i(indices, "indices")
        b, c, n = features.shape
        m = indices.shape[1]
        out = torch.empty(b, c, m, dtype=torch.float32, device=features.device)
        L.check(L.lib().bdm_gather_features_forward(b, c, n, m, L.ptr(features), L.ptr(indices), L.ptr(out),
                                                    L.stream()), "gather_features_forward")
        return out

    # ball_query.cpp:6-30
    @staticmethod
    def ball_query(centers_coords, points_coords, radius, num_neighbors):
        _chk_f(centers_coords, "centers_coords"); _chk_f(points_coords, "points_coords")
        b, _, m = centers_coords.shape
        n = points_coords.shape[2]
        out = torch.empty(b, m, int(num_neighbors), dtype=torch.int32, device=centers_coords.device)
        L.check(L.lib().bdm_ball_query(b, n, m, L.c_float(radius), int(num_neighbors), L.ptr(centers_coords),
                                       L.ptr(points_coords), L.ptr(out), L.stream()), "ball_query")
        return out

    # grouping.cpp:6-24
    @staticmethod
    def grouping_forward(features, indices):
        _chk_f(features, "features"); _chk_i(indices, "indices")
        b, c, n = features.shape
        _, m, u = indices.shape
        out = torch.empty(b, c, m, u, dtype=torch.float32, device=features.device)
        L.check(L.lib().bdm_grouping_forward(b, c, n, m, u, L.ptr(features), L.ptr(indices), L.ptr(out),
                                             L.stream()), "grouping_forward")
        return out

    # neighbor_interpolate.cpp:6-40
    @staticmethod
    def three_nearest_neighbors_interpolate_forward(points_coords, centers_coords, centers_features):
        _chk_f(points_coords, "points_coords"); _chk_f(centers_coords, "centers_coords")
        _chk_f(centers_features, "centers_features")
        b, c, m = centers_features.shape
        n = points_coords.shape[2]
        dev = points_coords.device
        idx = torch.empty(b, 3, n, dtype=torch.int32, device=dev)
        w = torch.empty(b, 3, n, dtype=torch.float32, device=dev)
        out = torch.empty(b, c, n, dtype=torch.float32, device=dev)
        L.check(L.lib().bdm_three_nn_interpolate_forward(b, c, m, n, L.ptr(points_coords), L.ptr(centers_coords),
                                                         L.ptr(centers_features), L.ptr(out), L.ptr(idx), L.ptr(w),
                                                         L.stream()), "three_nearest_neighbors_interpolate_forward")
        return [out, idx, w]

    # vox.cpp:17-43
    @staticmethod
    def avg_voxelize_forward(features, coords, resolution):
        _chk_f(features, "features"); _chk_i(coords, "coords")
        b, c, n = features.shape
        r = int(resolution)
        dev = features.device
        out = torch.empty(b, c, r ** 3, dtype=torch.float32, device=dev)
        ind = torch.empty(b, n, dtype=torch.int32, device=dev)
        cnt = torch.empty(b, r ** 3, dtype=torch.int32, device=dev)
        ws = torch.empty(L.lib().bdm_voxelize_workspace_bytes(b, n, r), dtype=torch.uint8, device=dev)
        L.check(L.lib().bdm_avg_voxelize_forward(b, c, n, r, L.ptr(features), L.ptr(coords), L.ptr(out), L.ptr(ind),
                                                 L.ptr(cnt), L.ptr(ws), L.stream()), "avg_voxelize_forward")
        return [out, ind, cnt]

    # trilinear_devox.cpp:18-55
    @staticmethod
    def trilinear_devoxelize_forward(resolution, is_training, coords, features):
        _chk_f(features, "features"); _chk_f(coords, "coords")
        b, c, _ = features.shape
        n = coords.shape[2]
        dev = features.device
        out = torch.empty(b, c, n, dtype=torch.float32, device=dev)
        if is_training:  # also saves the corner indices / weights for the backward pass
            inds = torch.empty(b, 8, n, dtype=torch.int32, device=dev)
            wgts = torch.empty(b, 8, n, dtype=torch.float32, device=dev)
            L.check(L.lib().bdm_trilinear_devoxelize_forward_training(b, c, n, int(resolution), L.ptr(coords), L.ptr(features),
                                                                      L.ptr(out), L.ptr(inds), L.ptr(wgts), L.stream()),
                    "trilinear_devoxelize_forward(training)")
            return [out, inds, wgts]
        L.check(L.lib().bdm_trilinear_devoxelize_forward(b, c, n, int(resolution), L.ptr(coords), L.ptr(features),
                                                         L.ptr(out), L.stream()), "trilinear_devoxelize_forward")
        # eval mode returns 1-element placeholders (trilinear_devox.cpp:45-53)
        return [out, torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)]

    # ---- training half (gradients): sampling.cpp:25-41, grouping.cpp:26-44, neighbor_interpolate.cpp:42-66,
    #      trilinear_devox.cpp:57-83, vox.cpp:45-69 -------------------------------------------------------------
    @staticmethod
    def gather_features_backward(grad_y, indices, n):
        _chk_f(grad_y, "grad_y"); _chk_i(indices, "indices")
        b, c, m = grad_y.shape
        gx = torch.empty(b, c, int(n), dtype=torch.float32, device=grad_y.device)
        L.check(L.lib().bdm_gather_features_backward(b, c, int(n), m, L.ptr(grad_y), L.ptr(indices), L.ptr(gx), L.stream()),
                "gather_features_backward")
        return gx

    @staticmethod
    def grouping_backward(grad_y, indices, n):
        _chk_f(grad_y, "grad_y"); _chk_i(indices, "indices")
        b, c, m, u = grad_y.shape
        gx = torch.empty(b, c, int(n), dtype=torch.float32, device=grad_y.device)
        L.check(L.lib().bdm_grouping_backward(b, c, int(n), m, u, L.ptr(grad_y), L.ptr(indices), L.ptr(gx), L.stream()),
                "grouping_backward")
        return gx

    @staticmethod
    def three_nearest_neighbors_interpolate_backward(grad_y, indices, weights, m):
        _chk_f(grad_y, "grad_y"); _chk_i(indices, "indices"); _chk_f(weights, "weights")
        b, c, n = grad_y.shape
        gx = torch.empty(b, c, int(m), dtype=torch.float32, device=grad_y.device)
        L.check(L.lib().bdm_three_nn_interpolate_backward(b, c, n, int(m), L.ptr(grad_y), L.ptr(indices), L.ptr(weights), L.ptr(gx),
                                                          L.stream()), "three_nearest_neighbors_interpolate_backward")
        return gx

    @staticmethod
    def trilinear_devoxelize_backward(grad_y, indices, weights, resolution):
        _chk_f(grad_y, "grad_y"); _chk_i(indices, "indices"); _chk_f(weights, "weights")
        b, c, n = grad_y.shape
        r = int(resolution)
        gx = torch.empty(b, c, r ** 3, dtype=torch.float32, device=grad_y.device)
        L.check(L.lib().bdm_trilinear_devoxelize_backward(b, c, n, r, L.ptr(indices), L.ptr(weights), L.ptr(grad_y), L.ptr(gx),
                                                          L.stream()), "trilinear_devoxelize_backward")
        return gx

    @staticmethod
    def avg_voxelize_backward(grad_y, indices, cnt):
        _chk_f(grad_y, "grad_y"); _chk_i(indices, "indices"); _chk_i(cnt, "cnt")
        b, c, s = grad_y.shape
        n = indices.shape[1]
        r = round(s ** (1.0 / 3.0))
        assert r ** 3 == s, "grad_y must be (B, C, r^3)"
        gx = torch.empty(b, c, n, dtype=torch.float32, device=grad_y.device)
        L.check(L.lib().bdm_avg_voxelize_backward(b, c, n, r, L.ptr(indices), L.ptr(cnt), L.ptr(grad_y), L.ptr(gx), L.stream()),
                "avg_voxelize_backward")
        return gx


_backend = _Backend()

__all__ = ["_backend"]
