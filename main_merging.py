"""BDM-Merging sampling entry point -- drop-in for the sampling half of experiments/main_merging.py
(run.job=sample_bdm_merging; outputs under sample_bdm_merging/{gt,pred}/<category>/).  The fusion-decoder
training job (run.job=training_bdm_merging) is out of scope for the MI355X sampling path."""
import sys
from pathlib import Path

import torch

from main_blending import build_models, get_dataloader, save_outputs


def main(argv=None):
    from bdm_amd.config import parse_overrides
    from bdm_amd.distributed import barrier, gpu_turn, init_from_env, shared_run_dir
    from bdm_amd.sampling import batch_streams, bdm_merging
    cfg = parse_overrides(sys.argv[1:] if argv is None else argv)
    rank, local_rank, world = init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(cfg.run.seed + rank)
    if cfg.run.job == "training_bdm_merging":
        raise NotImplementedError("fusion-decoder training is out of scope (sampling hot path only)")
    if cfg.run.job != "sample_bdm_merging":
        raise ValueError(f"Invalid job: {cfg.run.job}")
    out_root = Path(shared_run_dir(cfg, rank, world)) / "sample_bdm_merging"
    recon_model, prior_model, fusion_model = build_models(cfg, device, need_fusion=True)
    for batch_idx, batch in enumerate(get_dataloader(cfg, rank, world)):
        if cfg.run.num_sample_batches is not None and batch_idx >= cfg.run.num_sample_batches:
            break
        batch = batch.to(device)
        for sample_idx in range(cfg.run.num_samples):
            with gpu_turn(device):
                output = bdm_merging(None, batch, cfg, prior_model, recon_model, fusion_model,
                                     streams=batch_streams(cfg, batch, device, sample_idx))
            save_outputs(out_root, batch, output.points_padded(), sample_idx, cfg.run.num_samples)
    barrier()
    if rank == 0:
        print("Saved samples to:", out_root.absolute())
    return out_root


if __name__ == "__main__":
    main()
