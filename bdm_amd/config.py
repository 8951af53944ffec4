"""Structured configuration with the reference's group / key names and defaults
(experiments/config/structured.py:14-325) and Hydra-style `group.key=value` command-line overrides
(hydra-core / omegaconf are third-party and not installed here; only the override grammar the shell
recipes use -- example_sample_blending.sh:20-40 -- is re-implemented: `a.b=value`, `group=name`,
YAML scalars and lists as values)."""
import os
from dataclasses import asdict, dataclass, field, fields
from typing import Any, List, Optional

import yaml


@dataclass
class RunConfig:  # structured.py:14-56
    name: str = "debug"
    job: str = "train"
    mixed_precision: str = "fp16"
    cpu: bool = False
    seed: int = 42
    manual_seed: Optional[int] = None
    num_inference_steps: int = 1000
    diffusion_scheduler: Optional[str] = "ddpm"
    num_samples: int = 1
    num_sample_batches: Optional[int] = None
    sample_from_ema: bool = False
    sample_save_evolutions: bool = True
    freeze_feature_model: bool = True
    interact_timestep: Optional[List] = None
    max_fusion_steps: int = 20000
    save_dir: Optional[str] = None
    # not in the reference: "reference" = its draws (CPU generator for the initial cloud and the blend masks, the device's
    # global generator for DDPM / PVD noise, seeded seed + rank); "per_shape" = Philox streams keyed by (seed, global
    # shape index), identical results for any rank count / batch size (bdm_amd/rng.py)
    rng: str = "reference"


@dataclass
class AutomaticalPriorConfig:  # structured.py:58-64
    roll_step: Optional[int] = 16
    milestones: Optional[List] = None
    prior_ckpt: Optional[str] = None
    recon_ckpt: Optional[str] = None
    fusion_ckpt: Optional[str] = None


@dataclass
class LoggingConfig:  # structured.py:67-70 (wandb is not available offline: default off here)
    wandb: bool = False
    wandb_project: str = "bdm"


@dataclass
class PointCloudDiffusionModelConfig:  # structured.py:73-111
    image_size: int = 224
    image_feature_model: str = "vit_small_patch16_224_msn"
    use_local_colors: bool = True
    use_local_features: bool = True
    use_global_features: bool = False
    use_mask: bool = False
    use_distance_transform: bool = False
    scale_factor: float = 1.0
    colors_mean: float = 0.5
    colors_std: float = 0.5
    color_channels: int = 3
    predict_shape: bool = True
    predict_color: bool = False
    beta_start: float = 1e-5
    beta_end: float = 8e-3
    beta_schedule: str = "linear"
    point_cloud_model: str = "pvcnn"
    point_cloud_model_embed_dim: int = 64

    def as_kwargs(self):
        return asdict(self)


@dataclass
class PointCloudDatasetConfig:  # structured.py:127-140 (+ dataset-specific keys; `synthetic` is new)
    type: str = "synthetic"
    eval_split: str = "val"
    max_points: int = 16_384
    image_size: int = 224
    scale_factor: float = 1.0
    subset_ratio: float = 1.0
    restrict_model_ids: Optional[List] = None
    root: Optional[str] = None
    r2n2_dir: Optional[str] = None
    category: str = "chair"
    # ShapeNetR2N2Config / Pix3DConfig keys (structured.py:139-160)
    pc_dict: Optional[str] = None          # default: pc_dict_v2.json (R2N2) / pix3d.json (Pix3D)
    split_file: str = "R2N2_split.json"
    views_rel_path: str = "ShapeNetRendering"
    which_view_from24: str = "00"
    mask_images: bool = False
    start_ratio: float = 0.0
    processed: bool = True
    num_shapes: int = 16  # synthetic only


@dataclass
class DataloaderConfig:  # structured.py:176-179
    batch_size: int = 8
    num_workers: int = 6


@dataclass
class CheckpointConfig:  # structured.py:189-195
    resume: Optional[str] = None
    resume_training: bool = True


@dataclass
class ProjectConfig:  # structured.py:270-295 (sampling-relevant groups)
    run: RunConfig = field(default_factory=RunConfig)
    aux_run: AutomaticalPriorConfig = field(default_factory=AutomaticalPriorConfig)
    logging: LoggingConfig = field(default_factory=LoggingConfig)
    dataset: PointCloudDatasetConfig = field(default_factory=PointCloudDatasetConfig)
    dataloader: DataloaderConfig = field(default_factory=DataloaderConfig)
    model: PointCloudDiffusionModelConfig = field(default_factory=PointCloudDiffusionModelConfig)
    checkpoint: CheckpointConfig = field(default_factory=CheckpointConfig)


_IGNORED_GROUPS = {"optimizer", "scheduler", "ema", "loss", "augmentations", "hydra"}  # training-only groups


def parse_overrides(argv, cfg: Optional[ProjectConfig] = None) -> ProjectConfig:
    cfg = cfg or ProjectConfig()
    for arg in argv:
        if "=" not in arg:
            raise ValueError(f"expected key=value, got {arg!r}")
        key, raw = arg.split("=", 1)
        key = key.lstrip("+")
        value = yaml.safe_load(raw) if raw != "" else None
        if key == "dataset.which_view_from24" and value is not None:
            value = f"{int(value):02d}" if not isinstance(value, str) else value  # '00' must stay a string
        parts = key.split(".")
        if len(parts) == 1:
            if parts[0] == "dataset":  # config-group selection: dataset=shapenet_r2n2 | pix3d | synthetic
                cfg.dataset.type = str(value)
                continue
            if parts[0] in _IGNORED_GROUPS or parts[0] == "model":
                continue
            raise KeyError(f"unknown config group {parts[0]!r}")
        group, name = parts[0], parts[1]
        if group in _IGNORED_GROUPS:
            continue
        node = getattr(cfg, group, None)
        if node is None:
            raise KeyError(f"unknown config group {group!r}")
        if name not in {f.name for f in fields(node)}:
            if group in ("dataset", "logging"):  # dataset-specific keys of the real loaders (views_rel_path, ...)
                continue
            raise KeyError(f"unknown config key {key!r}")
        setattr(node, name, value)
    # interpolations of the reference's config: model.image_size = ${dataset.image_size}, model.scale_factor = ${dataset.scale_factor}
    cfg.model.image_size = cfg.dataset.image_size
    cfg.model.scale_factor = cfg.dataset.scale_factor
    return cfg


def run_dir(cfg: ProjectConfig) -> str:
    """${run.save_dir}/${run.name}/<timestamp> (structured.py:9-11); the CLI chdirs into it as Hydra does."""
    import datetime
    base = cfg.run.save_dir or "./outputs"
    return os.path.join(base, cfg.run.name, datetime.datetime.now().strftime("%Y-%m-%d--%H-%M-%S"))
