"""BDM-Blending sampling entry point -- drop-in for experiments/main_blending.py (same `group.key=value`
overrides, same job name, same output tree sample_bdm_blending/{gt,pred,images}/<category>/<name>.{ply,png} under
${run.save_dir}/${run.name}/<timestamp>).  One process per GPU; launch with
`python -m torch.distributed.run --nproc-per-node N main_blending.py ...` to shard batches over GPUs.

    python main_blending.py run.job=sample_bdm_blending dataset=synthetic dataset.max_points=4096 \
        dataloader.batch_size=16 aux_run.roll_step=16 aux_run.milestones=[1000,968,936,872,128,64,32,0]
"""
import os
import sys
from pathlib import Path

import torch


def build_models(cfg, device, need_fusion=False, need_prior=True):
    from bdm_amd.model import get_fusion_model, get_model
    from bdm_amd.pvd import prepare_pvd_model
    from bdm_amd.utils.procedural import fill_module_
    model = get_model(cfg)
    if cfg.checkpoint.resume:
        state = torch.load(cfg.checkpoint.resume, map_location="cpu")["model"]  # training_utils.py:283-290
        state = {k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in state.items()}
        print("load_state_dict:", model.load_state_dict(state, strict=False))
    else:
        print("checkpoint.resume not given: procedural random-init weights (benchmark mode)")
        fill_module_(model, seed=cfg.run.seed)
    model = model.to(device).eval()
    if not need_prior:  # vanilla PC^2 sampling (main.py run.job=sample)
        return model, None, None
    opt = {"model": cfg.aux_run.prior_ckpt, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}
    pvd_model = prepare_pvd_model(opt, device)
    fusion_model = None
    if need_fusion:
        fusion_model = get_fusion_model(cfg, pvd_model, model)
        if cfg.aux_run.fusion_ckpt:
            state = torch.load(cfg.aux_run.fusion_ckpt, map_location="cpu")["model"]
            print("fusion load_state_dict:", fusion_model.load_state_dict(state, strict=False))
        else:
            fill_module_(fusion_model.fusion_model.model.projs, seed=cfg.run.seed, prefix="projs.")
        fusion_model = fusion_model.to(device).eval()
    return model, pvd_model, fusion_model


def get_dataloader(cfg, rank, world):
    from bdm_amd.data import FrameData, SyntheticShapes
    from bdm_amd.distributed import shard_indices
    if cfg.dataset.type in ("shapenet_r2n2", "pix3d"):  # the reference's loaders (dataset/__init__.py:get_dataset)
        from bdm_amd.datasets import get_dataset
        _, val, _ = get_dataset(cfg, rank, world)
        return (FrameData(**b) for b in val)
    if cfg.dataset.type != "synthetic":
        raise NotImplementedError(f"dataset={cfg.dataset.type} (the BDM recipes use shapenet_r2n2 or pix3d; synthetic = benchmark inputs)")
    idx = shard_indices(cfg.dataset.num_shapes, rank, world)
    return SyntheticShapes(idx, cfg.dataloader.batch_size, seed=cfg.run.seed, image_size=cfg.dataset.image_size,
                           num_points=cfg.dataset.max_points, category=cfg.dataset.category)


def save_outputs(output_dir, batch, clouds, sample_idx, num_samples):
    """main_blending.py:413-455: {gt,pred}/<category>/<name>.ply and images/<category>/<name>.png."""
    from bdm_amd.io import save_image_png, save_pointcloud_ply
    for i in range(clouds.shape[0]):
        name, cat = batch.sequence_name[i], batch.sequence_category[i]
        stem = f"{name}-{sample_idx}" if num_samples > 1 else f"{name}"
        gt = batch.sequence_point_cloud[i]
        gt = gt.points_padded()[0] if hasattr(gt, "points_padded") else gt
        save_pointcloud_ply(gt.cpu().numpy(), output_dir / "gt" / cat / f"{stem}.ply")
        save_pointcloud_ply(clouds[i].cpu().numpy(), output_dir / "pred" / cat / f"{stem}.ply")
        save_image_png(batch.image_rgb[i].cpu().numpy(), output_dir / "images" / cat / f"{stem}.png")


def main(argv=None):
    from bdm_amd.config import parse_overrides
    from bdm_amd.distributed import barrier, gpu_turn, init_from_env, shared_run_dir
    from bdm_amd.sampling import batch_streams, bdm_blending
    cfg = parse_overrides(sys.argv[1:] if argv is None else argv)
    rank, local_rank, world = init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(cfg.run.seed + rank)  # training_utils.py:373-378 (set_seed: python, numpy and torch generators)
    import random as _random
    import numpy as _np
    _random.seed(cfg.run.seed + rank)
    _np.random.seed(cfg.run.seed + rank)
    if cfg.run.job != "sample_bdm_blending":
        raise ValueError(f"Invalid job: {cfg.run.job}")
    out_root = Path(shared_run_dir(cfg, rank, world)) / "sample_bdm_blending"
    model, pvd_model, _ = build_models(cfg, device)
    generator = torch.Generator().manual_seed(cfg.run.manual_seed) if cfg.run.manual_seed else None
    for batch_idx, batch in enumerate(get_dataloader(cfg, rank, world)):
        if cfg.run.num_sample_batches is not None and batch_idx >= cfg.run.num_sample_batches:
            break
        batch = batch.to(device)
        for sample_idx in range(cfg.run.num_samples):
            with gpu_turn(device):  # no-op unless several ranks share one GPU (test aid)
                output = bdm_blending(None, batch, cfg, model, pvd_model, generator=generator,
                                      streams=batch_streams(cfg, batch, device, sample_idx))
            save_outputs(out_root, batch, output.points_padded(), sample_idx, cfg.run.num_samples)
    barrier()
    if rank == 0:
        print("Saved samples to:", out_root.absolute())
    return out_root


if __name__ == "__main__":
    main()
