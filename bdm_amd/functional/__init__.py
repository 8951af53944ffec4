"""Reference-compatible functional API (experiments/model/pvcnn/modules/functional/__init__.py),
inference-only, on the HIP backend."""
from .backend import _backend


def ball_query(centers_coords, points_coords, radius, num_neighbors):
    """functional/ball_query.py:8-19"""
    return _backend.ball_query(centers_coords.contiguous(), points_coords.contiguous(), radius, num_neighbors)


def grouping(features, indices):
    """functional/grouping.py:10-24 (forward)"""
    return _backend.grouping_forward(features.contiguous(), indices.int().contiguous())


def gather(features, indices):
    """functional/sampling.py:11-25 (forward)"""
    return _backend.gather_features_forward(features.contiguous(), indices.int().contiguous())


def furthest_point_sample(coords, num_samples):
    """functional/sampling.py:37-48"""
    coords = coords.contiguous()
    return gather(coords, _backend.furthest_point_sampling(coords, num_samples))


def nearest_neighbor_interpolate(points_coords, centers_coords, centers_features):
    """functional/interpolatation.py:10-27 (forward)"""
    return _backend.three_nearest_neighbors_interpolate_forward(
        points_coords.contiguous(), centers_coords.contiguous(), centers_features.contiguous())[0]


def avg_voxelize(features, coords, resolution):
    """functional/voxelization.py:10-24 (forward)"""
    b, c, _ = features.shape
    out = _backend.avg_voxelize_forward(features.contiguous(), coords.int().contiguous(), resolution)[0]
    return out.view(b, c, resolution, resolution, resolution)


def trilinear_devoxelize(features, coords, resolution, is_training=False):
    """functional/devoxelization.py:10-27 (forward)"""
    B, C = features.shape[:2]
    return _backend.trilinear_devoxelize_forward(resolution, is_training, coords.contiguous(),
                                                 features.contiguous().view(B, C, -1))[0]


__all__ = ["ball_query", "grouping", "gather", "furthest_point_sample", "nearest_neighbor_interpolate",
           "avg_voxelize", "trilinear_devoxelize", "_backend"]
