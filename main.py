"""Vanilla PC^2 sampling entry point -- drop-in for the `run.job=sample` branch of experiments/main.py (main.py:148-160 ->
sample(), main.py:454-601; the recipe of example_sample.sh, BASELINE.json configs[0]).  Same `group.key=value` overrides; output
tree sample/{gt,pred,images,metadata,evolutions}/<category>/<sequence_name>.{ply,png,pth} under
${run.save_dir}/${run.name}/<timestamp>.  The other jobs of that file (train, vis, sample_bdm_*) are either out of scope for the
MI355X sampling path or live in main_blending.py / main_merging.py.

    python main.py run.job=sample dataset=synthetic dataset.max_points=1024 run.num_inference_steps=100 dataloader.batch_size=1
"""
import sys
from pathlib import Path

import torch

from main_blending import build_models, get_dataloader, save_outputs


def main(argv=None):
    from bdm_amd.config import parse_overrides
    from bdm_amd.distributed import barrier, gpu_turn, init_from_env, shared_run_dir
    cfg = parse_overrides(sys.argv[1:] if argv is None else argv)
    if cfg.run.job in ("train", "vis"):
        raise NotImplementedError(f"run.job={cfg.run.job} is out of scope (sampling hot path only)")
    if cfg.run.job != "sample":
        raise ValueError(f"Invalid job: {cfg.run.job}")
    rank, local_rank, world = init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(cfg.run.seed + rank)  # training_utils.py:373-378
    out_root = Path(shared_run_dir(cfg, rank, world)) / "sample"
    model, _, _ = build_models(cfg, device, need_prior=False)
    for batch_idx, batch in enumerate(get_dataloader(cfg, rank, world)):
        if cfg.run.num_sample_batches is not None and batch_idx >= cfg.run.num_sample_batches:
            break
        batch = batch.to(device)
        for sample_idx in range(cfg.run.num_samples):
            with gpu_turn(device):
                # main.py:497-504: model(batch, mode="sample", return_sample_every_n_steps=10, ...) -> (output, all_outputs)
                output, all_outputs = model(batch, mode="sample", num_points=cfg.dataset.max_points, return_sample_every_n_steps=10,
                                            scheduler=cfg.run.diffusion_scheduler, num_inference_steps=cfg.run.num_inference_steps,
                                            disable_tqdm=True)
            clouds = output.points_padded()
            save_outputs(out_root, batch, clouds, sample_idx, cfg.run.num_samples)
            for i in range(clouds.shape[0]):  # main.py:553-596: camera / bookkeeping per sample, optional evolutions
                name, cat = batch.sequence_name[i], batch.sequence_category[i]
                stem = f"{name}-{sample_idx}" if cfg.run.num_samples > 1 else f"{name}"
                for sub in ("metadata", "evolutions"):
                    (out_root / sub / cat).mkdir(parents=True, exist_ok=True)
                # the reference stores the BATCH's camera in every sample's metadata (experiments/main.py:577)
                cam = [c.to("cpu") if hasattr(c, "to") else c for c in batch.camera] if isinstance(batch.camera, (list, tuple)) else batch.camera.to("cpu")
                torch.save(dict(index=i, sequence_name=batch.sequence_name, sequence_category=batch.sequence_category,
                                frame_timestamp=batch.frame_timestamp, camera=cam,
                                image_size_hw=batch.image_size_hw, image_path=batch.image_path, depth_path=batch.depth_path,
                                mask_path=batch.mask_path, bbox_xywh=batch.bbox_xywh, crop_bbox_xywh=batch.crop_bbox_xywh,
                                sequence_point_cloud_path=batch.sequence_point_cloud_path, meta=batch.meta),
                           out_root / "metadata" / cat / f"{stem}.pth")
                if cfg.run.sample_save_evolutions:
                    # experiments/main.py:590-599 saves all_outputs[i]: ONE Pointclouds whose batch axis is the recorded steps
                    from bdm_amd.cameras import Pointclouds
                    torch.save(Pointclouds(points=torch.stack([o.points_padded()[i].cpu() for o in all_outputs])),
                               out_root / "evolutions" / cat / f"{stem}.pth")
    barrier()
    if rank == 0:
        print("Saved samples to:", out_root.absolute())
    return out_root


if __name__ == "__main__":
    main()
