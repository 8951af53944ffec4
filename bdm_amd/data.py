"""Batch container and synthetic inputs.

`FrameData` carries the fields of pytorch3d's implicitron FrameData that the sampling path reads
(dataset/shapenet_r2n2.py:508-536,601-612).  The real datasets are unavailable offline
(/root/reference/.MISSING_LARGE_BLOBS), so `SyntheticShapes` produces the benchmark inputs of
SURVEY.md 8d: uniform images, R2N2-style cameras looking at the origin, per-shape seeds keyed by the
GLOBAL shape index so that results do not depend on how shapes are sharded over ranks."""
from dataclasses import dataclass
from typing import Any, List, Optional

import torch

from .cameras import PerspectiveCameras, r2n2_camera


@dataclass
class FrameData:
    """The keys of the datasets' sample dict (dataset/shapenet_r2n2.py:508-536); the sampling path reads the first seven."""
    image_rgb: Optional[torch.Tensor] = None
    fg_probability: Optional[torch.Tensor] = None
    camera: Any = None
    sequence_point_cloud: Any = None
    sequence_name: Optional[List[str]] = None
    sequence_category: Optional[List[str]] = None
    frame_number: Optional[List[int]] = None
    frame_timestamp: Any = None
    image_size_hw: Any = None
    effective_image_size_hw: Any = None
    image_path: Any = None
    mask_crop: Any = None
    depth_path: Any = None
    depth_map: Any = None
    depth_mask: Any = None
    mask_path: Any = None
    bbox_xywh: Any = None
    crop_bbox_xywh: Any = None
    camera_quality_score: Any = None
    point_cloud_quality_score: Any = None
    sequence_point_cloud_path: Any = None
    sequence_point_cloud_idx: Any = None
    frame_type: Any = None
    meta: Any = None

    def to(self, device):
        import dataclasses
        cam = self.camera
        if isinstance(cam, (list, tuple)):
            cam = [c.to(device) for c in cam]
        elif cam is not None:
            cam = cam.to(device)
        fg = self.fg_probability.to(device) if torch.is_tensor(self.fg_probability) else self.fg_probability
        return dataclasses.replace(self, image_rgb=self.image_rgb.to(device), fg_probability=fg, camera=cam,
                                   sequence_point_cloud=None if self.sequence_point_cloud is None
                                   else self.sequence_point_cloud.to(device))

    def shape_indices(self):
        """Global shape indices of the batch (keys of the per-shape random streams): the dataset index when the loader
        recorded one (meta['dataset_index']), else frame_number (the synthetic shapes number themselves)."""
        if isinstance(self.meta, dict) and "dataset_index" in self.meta:
            return [int(v) for v in self.meta["dataset_index"]]
        return [int(v) for v in self.frame_number]


def shape_generator(seed: int, shape_index: int) -> torch.Generator:
    """One CPU generator per (seed, global shape index): rank-count-invariant noise streams."""
    return torch.Generator().manual_seed((int(seed) * 1_000_003 + int(shape_index) * 7919 + 12345) % (2 ** 63 - 1))


def synthetic_shape(seed: int, shape_index: int, image_size: int, num_points: int):
    g = shape_generator(seed, shape_index)
    image = torch.rand(3, image_size, image_size, generator=g)
    az = float(torch.rand(1, generator=g)) * 360.0
    el = 25.0 + float(torch.rand(1, generator=g)) * 5.0
    dist = 1.75 * (0.65 + float(torch.rand(1, generator=g)) * 0.30)
    gt = torch.randn(num_points, 3, generator=g) * 0.15
    return image, r2n2_camera(az, el, dist), gt


class SyntheticShapes:
    """Iterable of FrameData batches over global shape indices [first, first + count)."""

    def __init__(self, indices, batch_size, seed=42, image_size=224, num_points=4096, category="chair"):
        self.indices = list(indices)
        self.batch_size, self.seed, self.image_size, self.num_points, self.category = batch_size, seed, image_size, num_points, category

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, len(self.indices), self.batch_size):
            idx = self.indices[i:i + self.batch_size]
            items = [synthetic_shape(self.seed, j, self.image_size, self.num_points) for j in idx]
            yield FrameData(image_rgb=torch.stack([it[0] for it in items]), fg_probability=None,
                            camera=[it[1] for it in items], sequence_point_cloud=torch.stack([it[2] for it in items]),
                            sequence_name=[f"synthetic_{j:06d}" for j in idx], sequence_category=[self.category] * len(idx),
                            frame_number=list(idx))
