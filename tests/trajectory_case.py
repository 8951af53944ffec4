"""Shared set-up of the trajectory parity tests (tests/test_hip_teacher_forced.py, tests/test_hip_full_trajectory.py) and of
tools/chaos_probe.py: ONE description of a coupled BDM trajectory (weights, inputs, every random draw) that can be run by
the CPU oracle (oracle/ref_sampler.py) and by the HIP path (bdm_amd/sampling.py) on identical draws.

Random draws are keyed by (purpose, timestep), so the oracle looks them up by `t` while the HIP path receives them through
its replay hooks in the reference's program order (`program_order`), which follows main_blending.py:232-344 and
main_merging.py:415-520.

`head_scale` multiplies the LAST layer of each denoiser (classifier.2.weight/bias).  The reference initialises that layer
with N(0, 1e-6) (experiments/model/point_cloud_model.py:38-39): a freshly constructed reference model is therefore a
nearly-zero noise predictor, and procedural weights with a small head are the same regime.  With head_scale = 1 the
random-init network is chaotic (a 1-ulp input change is amplified to 2.6e-2 over 100 steps by flipping discrete decisions:
DESIGN.md section 5); tools/chaos_probe.py measures the oracle's own 1-ulp sensitivity as a function of head_scale.
"""
from types import SimpleNamespace

import torch

from helpers import seeded

MILESTONES = [1000, 968, 936, 872, 128, 64, 32, 0]


def program_order(milestones, roll_step, merging=False):
    """[(purpose, t), ...] in the order the reference's loops visit them.
    purposes: 'recon' main chain, 'branch' PC^2 branch of a window, 'prior' PVD branch, 'fuse' fused step (Merging)."""
    out = []
    times = len(milestones) - 1

    def seg(kind, start, end):
        out.extend((kind, t) for t in range(start - 1, end - 1, -1))

    for i in range(times):
        if i == 0:
            seg("recon", milestones[i], milestones[i + 1] - roll_step)
        elif i == times - 1:
            seg("recon", milestones[i] - roll_step, milestones[i + 1])
        else:
            seg("recon", milestones[i] - roll_step, milestones[i + 1])
            k = 1 if merging else 0
            seg("branch", milestones[i + 1], milestones[i + 1] - roll_step + k)
            seg("prior", milestones[i + 1], milestones[i + 1] - roll_step + k)
            if merging:
                out.append(("fuse", milestones[i + 1] - roll_step))
    return out


class KeyedNoise:
    """dict-like {t: tensor}: standard normal draws keyed by (base seed, t); `twin` repeats shape 0's draw for all shapes."""

    def __init__(self, shape, base, twin=False):
        self.shape, self.base, self.twin = tuple(shape), int(base), twin
        self._memo = {}

    def __getitem__(self, t):
        t = int(t)
        if t not in self._memo:
            if self.twin:
                one = seeded((1,) + self.shape[1:], self.base + t)
                self._memo[t] = one.expand(self.shape).contiguous()
            else:
                self._memo[t] = seeded(self.shape, self.base + t)
        return self._memo[t]

    def get(self, t, default=None):
        return self[t]


def scale_head_(module_sd_owner, key_prefix, scale):
    sd = module_sd_owner.state_dict()
    with torch.no_grad():
        for name in ("classifier.2.weight", "classifier.2.bias"):
            sd[key_prefix + name].mul_(scale)


def build(num_points, head_scale=1.0, milestones=None, roll_step=16, merging=False, twin=False, B=1, seed=3,
          head_scale_pvd=None):
    """twin=True: B = 2 where shape 1 is shape 0 with its initial cloud moved by +1 ulp (self-sensitivity probe)."""
    from bdm_amd.config import ProjectConfig
    from bdm_amd.data import FrameData, SyntheticShapes
    from bdm_amd.model import get_fusion_model, get_model
    from bdm_amd.pvd import prepare_pvd_model
    from bdm_amd.utils.procedural import fill_module_
    cfg = ProjectConfig()
    cfg.dataset.max_points = num_points
    cfg.aux_run.milestones, cfg.aux_run.roll_step = list(milestones or MILESTONES), roll_step
    model = fill_module_(get_model(cfg).eval(), seed=seed)
    scale_head_(model, "point_cloud_model.model.", head_scale)
    pvd = prepare_pvd_model({"model": f"procedural:{seed + 1}", "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cpu")
    scale_head_(pvd, "model.module.", head_scale if head_scale_pvd is None else head_scale_pvd)
    fusion = None
    if merging:
        fusion = get_fusion_model(cfg, pvd, model)  # the decoder / head copies are taken from the (scaled) PC^2 model
        fill_module_(fusion.fusion_model.model.projs, seed=seed + 2, prefix="projs.")  # non-zero "zero convs"
        fill_module_(fusion.feature_model, seed=seed, prefix="feature_model.")        # same image encoder as `model`
    nb = 1 if twin else B
    batch = next(iter(SyntheticShapes(range(nb), nb, seed=seed, image_size=224, num_points=num_points)))
    Bt = 2 if twin else B
    if twin:
        batch = FrameData(image_rgb=batch.image_rgb.repeat(2, 1, 1, 1), fg_probability=None, camera=[batch.camera[0]] * 2,
                          sequence_point_cloud=batch.sequence_point_cloud.repeat(2, 1, 1),
                          sequence_name=batch.sequence_name * 2, sequence_category=batch.sequence_category * 2,
                          frame_number=batch.frame_number * 2)
    init = seeded((nb, num_points, 3), 77 + seed)
    if twin:
        init = torch.cat([init, torch.nextafter(init, torch.full_like(init, float("inf")))], 0)
    c = SimpleNamespace(cfg=cfg, model=model, pvd=pvd, fusion=fusion, batch=batch, init=init, merging=merging, B=Bt,
                        N=num_points, milestones=cfg.aux_run.milestones, roll_step=roll_step)
    c.recon_noise = KeyedNoise((Bt, num_points, 3), 100_000, twin)
    c.branch_noise = KeyedNoise((Bt, num_points, 3), 200_000, twin)
    c.prior_noise = KeyedNoise((Bt, 3, num_points), 300_000, twin)
    c.fuse_noise = KeyedNoise((Bt, num_points, 3), 400_000, twin)
    n_masks = max(len(c.milestones) - 3, 0)
    c.masks = []
    for k in range(n_masks):
        m = torch.randint(0, 2, (1 if twin else Bt, num_points), generator=torch.Generator().manual_seed(900 + k))
        c.masks.append(m.expand(Bt, -1).contiguous() if twin else m)
    return c


def _cpu_sd(module):
    """state dict on the host (run_hip moves the modules to the GPU in place; the oracle may run after it)"""
    return {k: v.detach().cpu() for k, v in module.state_dict().items()}


def oracle_inputs(c):
    from bdm_amd.cameras import join_cameras
    from oracle import ref_vit
    local = ref_vit.local_conditioning(_cpu_sd(c.model), c.batch.image_rgb)
    cams = join_cameras(c.batch.camera).packed()
    return local, cams


def run_oracle(c, progress=False):
    """Final (B, N, 3) cloud of the CPU oracle."""
    from oracle import ref_sampler as R
    local, cams = oracle_inputs(c)
    if progress:
        import time
        t0, n = time.time(), [0]

        def trace(kind, t, x):
            n[0] += 1
            if n[0] % 50 == 0:
                msg = f"  [{time.time() - t0:6.0f} s] {n[0]:5d} steps, {kind} t={t}"
                if x.shape[0] == 2:
                    msg += f"  twin rel-L2 {float((x[1] - x[0]).norm() / x[0].norm()):.3e}"
                print(msg, flush=True)
        R.TRACE = trace
    try:
        if c.merging:
            from oracle import ref_vit
            local_f = ref_vit.local_conditioning(_cpu_sd(c.fusion), c.batch.image_rgb)
            return R.bdm_merging(_cpu_sd(c.model), _cpu_sd(c.pvd), _cpu_sd(c.fusion), c.init, cams, local, local_f,
                                 c.milestones, c.roll_step, c.recon_noise, c.branch_noise, c.prior_noise, c.fuse_noise)
        return R.bdm_blending(_cpu_sd(c.model), _cpu_sd(c.pvd), c.init, cams, local, c.milestones, c.roll_step,
                              c.recon_noise, c.branch_noise, c.prior_noise, c.masks)
    finally:
        R.TRACE = None


class segments:
    """with segments() as seg: run_hip(...) -> seg.clouds = the HIP path's (B, N, 3) host copies at the end of every schedule
    segment (bdm_amd.sampling.SEGMENT_HOOK), to be compared with the fixtures' `segment_k`."""

    def __enter__(self):
        from bdm_amd import sampling
        self.clouds = []
        sampling.SEGMENT_HOOK = lambda i, x: self.clouds.append(x.detach().cpu().clone())
        return self

    def __exit__(self, *exc):
        from bdm_amd import sampling
        sampling.SEGMENT_HOOK = None
        return False


def run_hip(c, device="cuda"):
    """Final (B, N, 3) cloud of the HIP path on the same draws, fed through the replay hooks in program order."""
    from bdm_amd.sampling import bdm_blending, bdm_merging
    order = program_order(c.milestones, c.roll_step, c.merging)
    pc2 = iter([(c.recon_noise if k == "recon" else c.branch_noise)[t] for k, t in order if k in ("recon", "branch") and t > 0])
    prior = iter([c.prior_noise[t] for k, t in order if k == "prior"])          # PVD draws at t == 0 too
    fuse = iter([c.fuse_noise[t] for k, t in order if k == "fuse" and t > 0])
    model, pvd = c.model.to(device), c.pvd.to(device)
    model.scheduler.noise_source = lambda shape, dev: next(pc2).to(dev)
    pvd.diffusion.noise_source = lambda shape, dev: next(prior).to(dev)
    try:
        if c.merging:
            fusion = c.fusion.to(device)
            fusion.scheduler.noise_source = lambda shape, dev: next(fuse).to(dev)
            try:
                out = bdm_merging(None, c.batch.to(device), c.cfg, pvd, model, fusion, init_noise=c.init)
            finally:
                fusion.scheduler.noise_source = None
        else:
            out = bdm_blending(None, c.batch.to(device), c.cfg, model, pvd, init_noise=c.init, blend_masks=c.masks)
    finally:
        model.scheduler.noise_source = None
        pvd.diffusion.noise_source = None
    for it, name in ((pc2, "PC^2"), (prior, "PVD"), (fuse, "fusion")):
        assert next(it, None) is None, f"{name} draws left over: program order differs from the sampler's"
    return out.points_padded().cpu()


# ---- one shape of a batch, and the per-shape Philox mode of the samplers ------------------------------------------------
class _Rows:
    """View of a KeyedNoise restricted to some shapes of the batch (same draws, rows `idx`)."""

    def __init__(self, full, idx):
        self.full, self.idx = full, list(idx)

    def __getitem__(self, t):
        return self.full[t][self.idx].contiguous()

    def get(self, t, default=None):
        return self[t]


def subset(c, idx):
    """The same case restricted to shapes `idx` of its batch: identical weights, inputs and draws for those shapes (every
    operator of the path is per-shape), so that the oracle -- or the HIP path at a smaller batch -- can re-run
    a sampled shape of a large batch."""
    import dataclasses
    idx = list(idx)
    b = c.batch
    batch = dataclasses.replace(b, image_rgb=b.image_rgb[idx].contiguous(), camera=[b.camera[i] for i in idx],
                                sequence_point_cloud=b.sequence_point_cloud[idx].contiguous(),
                                sequence_name=[b.sequence_name[i] for i in idx],
                                sequence_category=[b.sequence_category[i] for i in idx],
                                frame_number=[b.frame_number[i] for i in idx])
    s = SimpleNamespace(**vars(c))
    s.batch, s.B, s.init = batch, len(idx), c.init[idx].contiguous()
    for name in ("recon_noise", "branch_noise", "prior_noise", "fuse_noise"):
        setattr(s, name, _Rows(getattr(c, name), idx))
    s.masks = [m[idx].contiguous() for m in c.masks]
    return s


class PhiloxDraws:
    """dict-like {t: (1, ...) float32}: ONE shape's draws of the per-shape Philox mode (bdm_amd/rng.py), restated on the host by
    oracle/ref_rng.py.  `draw_of_t` maps a timestep to the draw index the sampler's stream counter has when it gets there."""

    def __init__(self, key, shape, purpose, draw_of_t):
        self.key, self.shape, self.purpose, self.draw_of_t = key, tuple(shape), purpose, dict(draw_of_t)

    def __getitem__(self, t):
        import numpy as np
        from oracle import ref_rng
        per = 1
        for s in self.shape:
            per *= s
        z = ref_rng.normal(self.key, per, self.draw_of_t[int(t)], self.purpose).astype(np.float32)
        return torch.from_numpy(z).reshape((1,) + self.shape)

    def get(self, t, default=None):
        return self[t]


def philox_shape_case(c, seed, shape_index, row):
    """The oracle-side description of shape `row` of case `c` (global index `shape_index`) when the HIP samplers draw from
    rng.ShapeStreams(seed, ...): initial cloud = INIT draw 0; DDPM noise = PC2 draws numbered in program order over the
    recon / branch steps with t > 0; PVD noise = PVD draws numbered over the prior steps (t == 0 included); masks = MASK
    draws; fused steps = FUSE draws."""
    import numpy as np
    from bdm_amd import rng as prod_rng
    from oracle import ref_rng
    key = ref_rng.shape_key(seed, shape_index)
    order = program_order(c.milestones, c.roll_step, c.merging)
    pc2_t = [t for k, t in order if k in ("recon", "branch") and t > 0]
    pvd_t = [t for k, t in order if k == "prior"]
    fuse_t = [t for k, t in order if k == "fuse" and t > 0]
    assert len(set(pc2_t)) == len(pc2_t) and len(set(pvd_t)) == len(pvd_t)
    s = subset(c, [row])
    N = c.N
    s.init = torch.from_numpy(ref_rng.normal(key, 3 * N, 0, prod_rng.INIT).astype(np.float32)).reshape(1, N, 3)
    pc2 = PhiloxDraws(key, (N, 3), prod_rng.PC2, {t: i for i, t in enumerate(pc2_t)})
    s.recon_noise = s.branch_noise = pc2
    s.prior_noise = PhiloxDraws(key, (3, N), prod_rng.PVD, {t: i for i, t in enumerate(pvd_t)})
    s.fuse_noise = PhiloxDraws(key, (N, 3), prod_rng.FUSE, {t: i for i, t in enumerate(fuse_t)})
    s.masks = [torch.from_numpy(ref_rng.bits(key, N, k, prod_rng.MASK)).reshape(1, N) for k in range(len(c.masks))]
    return s


def run_hip_streams(c, seed, shape_indices, device="cuda"):
    """Final (B, N, 3) cloud of the HIP samplers in the per-shape Philox mode (what bench.py runs): no injected draws."""
    from bdm_amd import rng as prod_rng
    from bdm_amd.sampling import bdm_blending, bdm_merging
    model, pvd = c.model.to(device), c.pvd.to(device)
    streams = prod_rng.ShapeStreams(seed, shape_indices, device)
    if c.merging:
        out = bdm_merging(None, c.batch.to(device), c.cfg, pvd, model, c.fusion.to(device), streams=streams)
    else:
        out = bdm_blending(None, c.batch.to(device), c.cfg, model, pvd, streams=streams)
    return out.points_padded().cpu()


# ---- C1: vanilla PC^2 sampling, one shape, N = 1024, 100 steps (model/model.py:123-214 of the reference) ----------------------------
def build_c1(head_scale=1.0, num_points=1024, steps=100, seed=11):
    """BASELINE.json configs[0] as a free-running trajectory with injected draws keyed by t (the set-up of
    tests/test_hip_trajectory.py's chaos monitor, with the head of the denoiser scaled like `build`)."""
    from bdm_amd.config import ProjectConfig
    from bdm_amd.data import SyntheticShapes
    from bdm_amd.model import get_model
    from bdm_amd.utils.procedural import fill_module_
    cfg = ProjectConfig()
    model = fill_module_(get_model(cfg).eval(), seed=seed)
    scale_head_(model, "point_cloud_model.model.", head_scale)
    batch = next(iter(SyntheticShapes(range(1), 1, seed=5, image_size=224, num_points=num_points)))
    stride = 1000 // steps
    ts = list(range(1000 - stride, -1, -stride))
    return SimpleNamespace(cfg=cfg, model=model, batch=batch, N=num_points, steps=steps, stride=stride, ts=ts,
                           noise={t: seeded((1, num_points, 3), 3000 + t) for t in ts}, init=seeded((1, num_points, 3), 99))


def run_oracle_c1(c, start=None, every=10):
    """-> (final (1, N, 3), [cloud after every `every`-th step]) of the CPU oracle (ref_sampler.RefDDPM, leading spacing)."""
    from bdm_amd.cameras import join_cameras
    from oracle import ref_net, ref_sampler as R, ref_vit
    sd = _cpu_sd(c.model)
    local = ref_vit.local_conditioning(sd, c.batch.image_rgb)
    cams = join_cameras(c.batch.camera).packed()
    ddpm = R.RefDDPM()
    x, snaps = (c.init if start is None else start).clone(), []
    for i, t in enumerate(c.ts):
        x_in = R.get_input_with_conditioning(x, cams, local)
        eps = ref_net.point_cloud_model_forward(sd, x_in, torch.full((1,), t), prefix="point_cloud_model.model.")
        x = ddpm.step(eps, t, x, c.noise[t] if t > 0 else None, prev_t=t - c.stride)
        if (i + 1) % every == 0:
            snaps.append(x.clone())
    return x, snaps


def run_hip_c1(c, device="cuda", every=10):
    """The HIP path on the same draws: the reverse loop in windows of `every` steps (each window one `_denoise_loop` call = the
    recorded step replayed, as `forward_sample` runs it) -> (final, [cloud after every window])."""
    model = c.model.to(device)
    sched = model.schedulers_map["ddpm"]
    sched.set_timesteps(c.steps)
    assert [int(v) for v in sched.timesteps] == c.ts
    it = iter([c.noise[t] for t in c.ts if t > 0])
    sched.noise_source = lambda shape, dev: next(it).to(dev)
    b = c.batch.to(device)
    y, snaps = c.init.to(device), []
    try:
        for k in range(0, len(c.ts), every):
            y = model._denoise_loop(y, b.camera, b.image_rgb, None, sched, c.ts[k:k + every])
            snaps.append(y.cpu())
    finally:
        sched.noise_source = None
    assert next(it, None) is None
    return y.cpu(), snaps
