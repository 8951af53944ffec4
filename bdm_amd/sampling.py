"""The BDM coupling schedules on the HIP path: `bdm_blending` (experiments/main_blending.py:186-347),
`bdm_merging` (experiments/main_merging.py:369-523) and `pvd_prior` (main_blending.py:175-183), with the
reference's argument lists (the `accelerator` argument is accepted and ignored: one process per GPU).

Segment arithmetic for milestones [1000,968,936,872,128,64,32,0], roll_step 16:
  Blending: 1000 PC^2 steps + 80 PVD steps + 5 per-point Bernoulli(1/2) selections per trajectory;
  Merging : 995 PC^2 + 75 PVD + 5 fused steps (PVCNN_fuse) per trajectory.
RNG placement follows the reference: the initial cloud is drawn on the CPU and moved
(main_blending.py:228), DDPM / PVD noise is drawn on the device, blend masks come from the CPU
generator (main_blending.py:330-338).  `noise` (optional) supplies every draw instead -- the replay
mode used by the parity tests.
"""
import torch

from . import _lib as L
from .cameras import Pointclouds
from .pvd import generate_pvd_xyz

DEFAULT_MILESTONES = [1000, 968, 936, 872, 128, 64, 32, 0]


@torch.no_grad()
def pvd_prior(pvd_model, points, start_time, end_time):
    """(B, N, 3) -> PVD steps t = start_time-1 ... end_time -> (B, N, 3)."""
    from . import ops
    pts = ops.transpose12(points.float())
    out = generate_pvd_xyz(pvd_model, pts, start_time, end_time)
    return ops.transpose12(out)


def center_points_(x):
    """x -= x.mean(dim=1, keepdim=True) on (B, N, 3), in place (main_blending.py:229)."""
    assert x.is_contiguous() and x.dim() == 3 and x.shape[2] == 3
    L.check(L.lib().bdm_center_points(x.shape[0], x.shape[1], L.ptr(x), L.stream()), "center_points")
    return x


def blend_select(recon, prior, indices):
    """pred[b, n] = (recon, prior)[indices[b, n]] (main_blending.py:326-344)."""
    recon, prior = recon.contiguous(), prior.contiguous()
    mask = indices.to(device=recon.device, dtype=torch.int64).contiguous()
    out = torch.empty_like(recon)
    L.check(L.lib().bdm_blend_select(L.c_ll(mask.numel()), L.ptr(recon), L.ptr(prior), L.ptr(mask), L.ptr(out), L.stream()),
            "blend_select")
    return out


def _schedule(cfg):
    """(roll_step, milestones, prior_roll_step, prior_milestones, times) -- main_blending.py:207-221: with DDIM the recon
    model runs on a 64-step grid while the PVD prior keeps its 1000-step DDPM chain (x16 roll, milestones * 1000 / 64)."""
    roll_step = cfg.aux_run.roll_step
    milestones = list(cfg.aux_run.milestones or DEFAULT_MILESTONES)
    if cfg.run.diffusion_scheduler == "ddim":
        prior_roll_step = int(roll_step * 16)
        prior_milestones = [int(i / 64 * 1000) for i in milestones]
    elif cfg.run.diffusion_scheduler == "ddpm":
        prior_roll_step, prior_milestones = roll_step, milestones
    else:
        raise NotImplementedError(f"scheduler {cfg.run.diffusion_scheduler!r}")
    return roll_step, milestones, prior_roll_step, prior_milestones, len(milestones) - 1


def _initial_cloud(B, num_points, device, init_noise=None):
    x = torch.randn(B, num_points, 3) if init_noise is None else init_noise.clone()
    return center_points_(x.to(device).contiguous())


@torch.no_grad()
def bdm_blending(accelerator, batch, cfg, model, pvd_model, generator=None, init_noise=None, blend_masks=None):
    """main_blending.py:186-347.  Returns Pointclouds of (B, N, 3)."""
    img, mask, camera = batch.image_rgb, batch.fg_probability, batch.camera
    roll_step, milestones, prior_roll_step, prior_milestones, times = _schedule(cfg)
    B, num_points = img.shape[0], cfg.dataset.max_points
    device = model.point_cloud_model.device
    common = dict(scheduler=cfg.run.diffusion_scheduler, num_inference_steps=cfg.run.num_inference_steps, disable_tqdm=True)
    pred_pc = _initial_cloud(B, num_points, device, init_noise)
    blends = 0
    for i in range(times):
        if i == 0:
            pred_pc = model.interaction_sample(pred_pc, camera, img, mask, start_time=milestones[i],
                                               end_time=milestones[i + 1] - roll_step, **common)
        elif i == times - 1:
            pred_pc = model.interaction_sample(pred_pc, camera, img, mask, start_time=milestones[i] - roll_step,
                                               end_time=milestones[i + 1], **common)
        else:
            pred_pc = model.interaction_sample(pred_pc, camera, img, mask, start_time=milestones[i] - roll_step,
                                               end_time=milestones[i + 1], **common)
            # Branch 1: reconstruction model, roll_step steps
            out_recon = model.interaction_sample(pred_pc.clone(), camera, img, mask, start_time=milestones[i + 1],
                                                 end_time=milestones[i + 1] - roll_step, **common)
            # Branch 2: prior model, roll_step steps from the same cloud
            out_prior = pvd_prior(pvd_model, pred_pc.clone(), start_time=prior_milestones[i + 1],
                                  end_time=prior_milestones[i + 1] - prior_roll_step)
            if blend_masks is not None:
                indices = blend_masks[blends]
            else:
                indices = torch.randint(0, 2, (B, num_points), generator=generator).long()
            blends += 1
            pred_pc = blend_select(out_recon, out_prior, indices)
    return Pointclouds(pred_pc)


@torch.no_grad()
def bdm_merging(accelerator, batch, cfg, prior_model, recon_model, fusion_model, init_noise=None):
    """main_merging.py:369-523.  Returns Pointclouds of (B, N, 3)."""
    img, mask, camera = batch.image_rgb, batch.fg_probability, batch.camera
    roll_step, milestones, prior_roll_step, prior_milestones, times = _schedule(cfg)
    B, num_points = img.shape[0], cfg.dataset.max_points
    device = recon_model.point_cloud_model.device
    common = dict(scheduler=cfg.run.diffusion_scheduler, num_inference_steps=cfg.run.num_inference_steps, disable_tqdm=True)
    pred_pc = _initial_cloud(B, num_points, device, init_noise)
    for i in range(times):
        if i == 0:
            pred_pc = recon_model.interaction_sample(pred_pc, camera, img, mask, start_time=milestones[i],
                                                     end_time=milestones[i + 1] - roll_step, **common)
        elif i == times - 1:
            pred_pc = recon_model.interaction_sample(pred_pc, camera, img, mask, start_time=milestones[i] - roll_step,
                                                     end_time=milestones[i + 1], **common)
        else:
            pred_pc = recon_model.interaction_sample(pred_pc, camera, img, mask, start_time=milestones[i] - roll_step,
                                                     end_time=milestones[i + 1], **common)
            out_recon = recon_model.interaction_sample(pred_pc.clone(), camera, img, mask, start_time=milestones[i + 1],
                                                       end_time=milestones[i + 1] - roll_step + 1, **common)
            out_prior = pvd_prior(prior_model, pred_pc.clone(), start_time=prior_milestones[i + 1],
                                  end_time=prior_milestones[i + 1] - prior_roll_step + 1)
            pred_pc = fusion_model.nstep_fuse(out_prior.contiguous(), out_recon.contiguous(), camera, img, mask,
                                              scheduler=cfg.run.diffusion_scheduler,
                                              num_inference_steps=cfg.run.num_inference_steps,
                                              timestep=milestones[i + 1] - roll_step)
    return Pointclouds(pred_pc)


def count_forwards(milestones=None, roll_step=16, merging=False):
    """(PC^2 forwards, PVD forwards, fusion forwards) per trajectory -- used by bench.py and the tests."""
    ms = list(milestones or DEFAULT_MILESTONES)
    times = len(ms) - 1
    pc2 = pvd = fuse = 0
    for i in range(times):
        if i == 0:
            pc2 += ms[i] - (ms[i + 1] - roll_step)
        elif i == times - 1:
            pc2 += (ms[i] - roll_step) - ms[i + 1]
        else:
            pc2 += (ms[i] - roll_step) - ms[i + 1]
            branch = roll_step - 1 if merging else roll_step
            pc2 += branch
            pvd += branch
            fuse += 1 if merging else 0
    return pc2, pvd, fuse
