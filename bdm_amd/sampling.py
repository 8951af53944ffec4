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
import contextlib

import torch

from . import _lib as L
from . import ops, rng
from .cameras import Pointclouds
from .pvd import generate_pvd_xyz

DEFAULT_MILESTONES = [1000, 968, 936, 872, 128, 64, 32, 0]

# optional callable(segment index, cloud (B, N, 3) on the device) invoked at the end of every schedule segment of bdm_blending /
# bdm_merging (diagnosis: where along a trajectory two implementations part; the parity tests set it, nothing else does)
SEGMENT_HOOK = None


@torch.no_grad()
def pvd_prior(pvd_model, points, start_time, end_time):
    """(B, N, 3) -> PVD steps t = start_time-1 ... end_time -> (B, N, 3)."""
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


def _initial_cloud(B, num_points, device, init_noise=None, streams=None):
    if init_noise is not None:
        x = init_noise.clone()
    elif streams is not None:
        x = streams.normal((B, num_points, 3), rng.INIT)
    else:
        x = torch.randn(B, num_points, 3)  # on the CPU, then moved (main_blending.py:228)
    return center_points_(x.to(device).contiguous())


@contextlib.contextmanager
def _streams_on(streams, *owners):
    """Attach per-shape noise streams to scheduler-like objects (`.streams`) for the duration of one trajectory.
    owners: (object, purpose or None) pairs."""
    if streams is None:
        yield
        return
    saved = []
    for obj, purpose in owners:
        saved.append((obj, obj.streams, getattr(obj, "stream_purpose", None)))
        obj.streams = streams
        if purpose is not None:
            obj.stream_purpose = purpose
    try:
        yield
    finally:
        for obj, st, purpose in saved:
            obj.streams = st
            if purpose is not None:
                obj.stream_purpose = purpose


def batch_streams(cfg, batch, device, sample_idx=0):
    """rng.ShapeStreams of a batch keyed by (run.seed, global shape index = batch.frame_number), or None when the
    configuration asks for the reference's global-generator draws (run.rng = "reference")."""
    if getattr(cfg.run, "rng", "reference") != "per_shape":
        return None
    # run.num_samples > 1: sample k of a shape uses the stream family seed + k * 1000003
    return rng.ShapeStreams(cfg.run.seed + 1000003 * int(sample_idx), batch.shape_indices(), device)


@torch.no_grad()
def bdm_blending(accelerator, batch, cfg, model, pvd_model, generator=None, init_noise=None, blend_masks=None, streams=None):
    """main_blending.py:186-347.  Returns Pointclouds of (B, N, 3).
    streams: rng.ShapeStreams -> every draw (initial cloud, DDPM / PVD noise, blend masks) comes from the shapes' own
    Philox streams (rank-count invariant); default: the reference's generators (or `run.rng=per_shape` in cfg)."""
    device = model.point_cloud_model.device
    streams = batch_streams(cfg, batch, device) if streams is None else streams
    sched = model.schedulers_map[cfg.run.diffusion_scheduler]
    with _streams_on(streams, (sched, rng.PC2), (pvd_model.diffusion, None)):
        out = _bdm_blending(batch, cfg, model, pvd_model, generator, init_noise, blend_masks, streams)
    ops.poll_h2_saturation()  # fp16x3 accuracy guard: one host check per trajectory
    return out


def _bdm_blending(batch, cfg, model, pvd_model, generator, init_noise, blend_masks, streams):
    img, mask, camera = batch.image_rgb, batch.fg_probability, batch.camera
    roll_step, milestones, prior_roll_step, prior_milestones, times = _schedule(cfg)
    B, num_points = img.shape[0], cfg.dataset.max_points
    device = model.point_cloud_model.device
    common = dict(scheduler=cfg.run.diffusion_scheduler, num_inference_steps=cfg.run.num_inference_steps, disable_tqdm=True)
    pred_pc = _initial_cloud(B, num_points, device, init_noise, streams)
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
            elif streams is not None:
                indices = streams.bits((B, num_points), rng.MASK)
            else:
                indices = torch.randint(0, 2, (B, num_points), generator=generator).long()
            blends += 1
            pred_pc = blend_select(out_recon, out_prior, indices)
        if SEGMENT_HOOK is not None:
            SEGMENT_HOOK(i, pred_pc)
    return Pointclouds(pred_pc)


@torch.no_grad()
def bdm_merging(accelerator, batch, cfg, prior_model, recon_model, fusion_model, init_noise=None, streams=None):
    """main_merging.py:369-523.  Returns Pointclouds of (B, N, 3).  streams: see bdm_blending."""
    device = recon_model.point_cloud_model.device
    streams = batch_streams(cfg, batch, device) if streams is None else streams
    name = cfg.run.diffusion_scheduler
    with _streams_on(streams, (recon_model.schedulers_map[name], rng.PC2), (fusion_model.schedulers_map[name], rng.FUSE),
                     (prior_model.diffusion, None)):
        out = _bdm_merging(batch, cfg, prior_model, recon_model, fusion_model, init_noise, streams)
    ops.poll_h2_saturation()
    return out


def _bdm_merging(batch, cfg, prior_model, recon_model, fusion_model, init_noise, streams):
    img, mask, camera = batch.image_rgb, batch.fg_probability, batch.camera
    roll_step, milestones, prior_roll_step, prior_milestones, times = _schedule(cfg)
    B, num_points = img.shape[0], cfg.dataset.max_points
    device = recon_model.point_cloud_model.device
    common = dict(scheduler=cfg.run.diffusion_scheduler, num_inference_steps=cfg.run.num_inference_steps, disable_tqdm=True)
    pred_pc = _initial_cloud(B, num_points, device, init_noise, streams)
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
        if SEGMENT_HOOK is not None:
            SEGMENT_HOOK(i, pred_pc)
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
