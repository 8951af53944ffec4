"""Reference-compatible functional API (experiments/model/pvcnn/modules/functional/__init__.py) on the HIP backend.
Each operator is a torch.autograd.Function over the `_backend` pair (forward, backward), exactly as the reference's
functional/{ball_query,grouping,sampling,interpolatation,voxelization,devoxelization}.py; under torch.no_grad() (the whole
sampling path) only the forward halves run."""
import torch
from torch.autograd import Function

from .backend import _backend


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    """functional/ball_query.py:8-19"""
    return _backend.ball_query(centers_coords.contiguous(), points_coords.contiguous(), radius, num_neighbors)


class Grouping(Function):
    """functional/grouping.py:8-30"""

    @staticmethod
    def forward(ctx, features, indices):
        features, indices = features.contiguous(), indices.int().contiguous()
        ctx.save_for_backward(indices)
        ctx.num_points = features.size(-1)
        return _backend.grouping_forward(features, indices)

    @staticmethod
    def backward(ctx, grad_output):
        indices, = ctx.saved_tensors
        return _backend.grouping_backward(grad_output.contiguous(), indices, ctx.num_points), None


grouping = Grouping.apply


class Gather(Function):
    """functional/sampling.py:8-32"""

    @staticmethod
    def forward(ctx, features, indices):
        features, indices = features.contiguous(), indices.int().contiguous()
        ctx.save_for_backward(indices)
        ctx.num_points = features.size(-1)
        return _backend.gather_features_forward(features, indices)

    @staticmethod
    def backward(ctx, grad_output):
        indices, = ctx.saved_tensors
        return _backend.gather_features_backward(grad_output.contiguous(), indices, ctx.num_points), None


gather = Gather.apply


def furthest_point_sample(coords, num_samples):
    """functional/sampling.py:37-48"""
    coords = coords.contiguous()
    return gather(coords, _backend.furthest_point_sampling(coords, num_samples))


class NeighborInterpolation(Function):
    """functional/interpolatation.py:8-33"""

    @staticmethod
    def forward(ctx, points_coords, centers_coords, centers_features):
        out, indices, weights = _backend.three_nearest_neighbors_interpolate_forward(
            points_coords.contiguous(), centers_coords.contiguous(), centers_features.contiguous())
        ctx.save_for_backward(indices, weights)
        ctx.num_centers = centers_coords.size(-1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        indices, weights = ctx.saved_tensors
        gx = _backend.three_nearest_neighbors_interpolate_backward(grad_output.contiguous(), indices, weights, ctx.num_centers)
        return None, None, gx


nearest_neighbor_interpolate = NeighborInterpolation.apply


class AvgVoxelization(Function):
    """functional/voxelization.py:8-37"""

    @staticmethod
    def forward(ctx, features, coords, resolution):
        features, coords = features.contiguous(), coords.int().contiguous()
        b, c, _ = features.shape
        out, indices, counts = _backend.avg_voxelize_forward(features, coords, resolution)
        ctx.save_for_backward(indices, counts)
        return out.view(b, c, resolution, resolution, resolution)

    @staticmethod
    def backward(ctx, grad_output):
        b, c = grad_output.shape[:2]
        indices, counts = ctx.saved_tensors
        gx = _backend.avg_voxelize_backward(grad_output.contiguous().view(b, c, -1), indices, counts)
        return gx, None, None


avg_voxelize = AvgVoxelization.apply


class TrilinearDevoxelization(Function):
    """functional/devoxelization.py:8-41"""

    @staticmethod
    def forward(ctx, features, coords, resolution, is_training=True):
        B, C = features.shape[:2]
        features = features.contiguous().view(B, C, -1)
        coords = coords.contiguous()
        outs, inds, wgts = _backend.trilinear_devoxelize_forward(resolution, is_training, coords, features)
        if is_training:
            ctx.save_for_backward(inds, wgts)
            ctx.r = resolution
        return outs

    @staticmethod
    def backward(ctx, grad_output):
        inds, wgts = ctx.saved_tensors
        gx = _backend.trilinear_devoxelize_backward(grad_output.contiguous(), inds, wgts, ctx.r)
        return gx.view(grad_output.size(0), grad_output.size(1), ctx.r, ctx.r, ctx.r), None, None, None


trilinear_devoxelize = TrilinearDevoxelization.apply  # as the reference (devoxelization.py:44): is_training defaults to True


__all__ = ["ball_query", "grouping", "gather", "furthest_point_sample", "nearest_neighbor_interpolate",
           "avg_voxelize", "trilinear_devoxelize", "_backend"]
