"""oracle/ref_sampler.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (torch fp32 / numpy) of the sampling loop around the denoisers:
  * DDPM step of diffusers 0.21.0 as called by experiments/model/model.py:193,286,563
    ("parity unpinned": diffusers is a third-party dependency absent from /root/reference and from
    this image; restated from its published algorithm, checked against the closed-form posterior);
  * PVD p_sample (experiments/pvd/__init__.py:136-224) -- pinned by tests/golden/pvd_gaussian_diffusion.npz;
  * projection conditioning (experiments/model/projection_model.py:127-157,179-231) with a brute-force
    restatement of pytorch3d's naive PointsRasterizer ("parity unpinned": pytorch3d absent);
  * coupling schedules (experiments/main_blending.py:232-344, experiments/main_merging.py:415-520).
All random draws are INJECTED (noise replay) so that the HIP path can be compared on
identical noise.
"""
import numpy as np
import torch

from . import ref_net


# ---------------------------------------------------------------------------------------------
# schedulers
# ---------------------------------------------------------------------------------------------
class RefDDPM:
    def __init__(self, beta_start=1e-5, beta_end=8e-3, T=1000):
        self.T = T
        self.betas = torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)

    def step(self, eps, t, x, z, prev_t=None):
        ac = self.alphas_cumprod
        prev_t = t - 1 if prev_t is None else prev_t
        abar_t = ac[t]
        abar_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
        beta_prod_t, beta_prod_prev = 1 - abar_t, 1 - abar_prev
        cur_alpha = abar_t / abar_prev
        cur_beta = 1 - cur_alpha
        x0 = (x - beta_prod_t ** 0.5 * eps) / abar_t ** 0.5
        prev = (abar_prev ** 0.5 * cur_beta) / beta_prod_t * x0 + cur_alpha ** 0.5 * beta_prod_prev / beta_prod_t * x
        if t > 0:
            var = torch.clamp((1 - abar_prev) / (1 - abar_t) * cur_beta, min=1e-20)
            prev = prev + (var ** 0.5) * z
        return prev


class RefPNDM:
    """diffusers 0.21.0 PNDMScheduler (skip_prk_steps=False, set_alpha_to_one=False, epsilon prediction, leading spacing),
    restated with torch CPU ops from the published algorithm ("parity unpinned": diffusers absent).  The reference exposes
    it as schedulers_map['pndm'] (experiments/model/model.py:61)."""

    def __init__(self, beta_start=1e-5, beta_end=8e-3, T=1000):
        self.T = T
        self.ac = torch.cumprod(1.0 - torch.linspace(beta_start, beta_end, T, dtype=torch.float32), dim=0)
        self.final = self.ac[0]

    def set_timesteps(self, n):
        self.n, ratio = n, self.T // n
        base = (np.arange(0, n) * ratio).round().astype(np.int64)
        prk = np.array(base[-4:]).repeat(2) + np.tile(np.array([0, ratio // 2]), 4)
        self.prk = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
        self.plms = base[:-3][::-1].copy()
        self.timesteps = np.concatenate([self.prk, self.plms]).astype(np.int64)
        self.cur_model_output, self.counter, self.cur_sample, self.ets = 0, 0, None, []

    def _prev(self, sample, t, prev_t, e):
        a_t = self.ac[t]
        a_prev = self.ac[prev_t] if prev_t >= 0 else self.final
        coeff = (a_prev / a_t) ** 0.5
        denom = a_t * (1 - a_prev) ** 0.5 + (a_t * (1 - a_t) * a_prev) ** 0.5
        return coeff * sample - (a_prev - a_t) * e / denom

    def step(self, e, t, sample):
        ratio = self.T // self.n
        if self.counter < len(self.prk):
            prev_t = t - (0 if self.counter % 2 else ratio // 2)
            t = int(self.prk[self.counter // 4 * 4])
            if self.counter % 4 == 0:
                self.cur_model_output = self.cur_model_output + 1 / 6 * e
                self.ets.append(e)
                self.cur_sample = sample
            elif (self.counter - 1) % 4 == 0 or (self.counter - 2) % 4 == 0:
                self.cur_model_output = self.cur_model_output + 1 / 3 * e
            else:
                e = self.cur_model_output + 1 / 6 * e
                self.cur_model_output = 0
            cur = self.cur_sample if self.cur_sample is not None else sample
            out = self._prev(cur, t, prev_t, e)
        else:
            prev_t = t - ratio
            if self.counter != 1:
                self.ets = self.ets[-3:]
                self.ets.append(e)
            else:
                prev_t, t = t, t + ratio
            if len(self.ets) == 1 and self.counter == 0:
                self.cur_sample = sample
            elif len(self.ets) == 1 and self.counter == 1:
                e = (e + self.ets[-1]) / 2
                sample, self.cur_sample = self.cur_sample, None
            elif len(self.ets) == 2:
                e = (3 * self.ets[-1] - self.ets[-2]) / 2
            elif len(self.ets) == 3:
                e = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
            else:
                e = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
            out = self._prev(sample, t, prev_t, e)
        self.counter += 1
        return out


class RefPVDDiffusion:
    """pvd/__init__.py:24-68 tables + :196-224 step."""

    def __init__(self):
        betas = np.linspace(0.0001, 0.02, 1000).astype(np.float64)
        alphas = 1.0 - betas
        acp = torch.from_numpy(np.cumprod(alphas, axis=0)).float()
        acp_prev = torch.from_numpy(np.append(1.0, acp[:-1])).float()
        b32, a32 = torch.from_numpy(betas).float(), torch.from_numpy(alphas).float()
        self.sqrt_recip = torch.sqrt(1.0 / acp).float()
        self.sqrt_recipm1 = torch.sqrt(1.0 / acp - 1).float()
        pv = b32 * (1.0 - acp_prev) / (1.0 - acp)
        self.logvar = torch.log(torch.max(pv, 1e-20 * torch.ones_like(pv)))
        self.c1 = b32 * torch.sqrt(acp_prev) / (1.0 - acp)
        self.c2 = (1.0 - acp_prev) * torch.sqrt(a32) / (1.0 - acp)

    def step(self, eps, t, x, z):
        x0 = self.sqrt_recip[t] * x - self.sqrt_recipm1[t] * eps
        mean = self.c1[t] * x0 + self.c2[t] * x
        mask = 0.0 if t == 0 else 1.0
        return mean + mask * torch.exp(0.5 * self.logvar[t]) * z


# ---------------------------------------------------------------------------------------------
# projection conditioning
# ---------------------------------------------------------------------------------------------
def project_points(points, cam):
    """points (N,3), cam (16,) = R row-major, T, focal, pp -> ndc x, y and view depth (float32)."""
    R, T, f, p = cam[:9].view(3, 3), cam[9:12], cam[12:14], cam[14:16]
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    xv = x * R[0, 0] + y * R[1, 0] + z * R[2, 0] + T[0]
    yv = x * R[0, 1] + y * R[1, 1] + z * R[2, 1] + T[1]
    zv = x * R[0, 2] + y * R[1, 2] + z * R[2, 2] + T[2]
    return f[0] * xv / zv + p[0], f[1] * yv / zv + p[1], zv


def rasterize_bruteforce(points, cam, H, W, radius):
    """pytorch3d naive PointsRasterizer, points_per_pixel = 1: (H, W) int64 index image, -1 = empty.
    Every pixel tests EVERY point: dx^2 + dy^2 < r^2 and z >= 0; nearest z wins, earliest index on ties."""
    u, v, d = project_points(points, cam)
    r2 = np.float32(radius) * np.float32(radius)
    idx = torch.full((H, W), -1, dtype=torch.int64)
    xs = 1.0 - (2.0 * torch.arange(W, dtype=torch.float32) + 1.0) / W
    valid = d >= 0
    for yi in range(H):
        yf = 1.0 - (2.0 * yi + 1.0) / H
        dy = yf - v
        dx = xs[:, None] - u[None, :]
        hit = (dx * dx + (dy * dy)[None, :] < r2) & valid[None, :]
        zz = torch.where(hit, d[None, :].expand(W, -1), torch.full((), float("inf")))
        best = zz.min(dim=1)
        # earliest index among equal z: argmin over (z, index)
        first = (zz == best.values[:, None]).float().argmax(dim=1)
        idx[yi] = torch.where(torch.isfinite(best.values), first, torch.full_like(first, -1))
    return idx


def rasterize_windowed(points, cam, H, W, radius, win=2):
    """Same index image as rasterize_bruteforce, visiting for every point only the pixels within +-win of its nearest
    pixel (valid while radius < win pixel pitches; equality with the brute-force version is a CPU test).  Used for
    long trajectories where the brute-force form would dominate the oracle's run time."""
    u, v, d = project_points(points, cam)
    r2 = np.float32(radius) * np.float32(radius)
    n = points.shape[0]
    xc = torch.nan_to_num(torch.round(((1.0 - u) * W - 1.0) * 0.5), nan=-4.0).clamp(-4, W + 4).long()
    yc = torch.nan_to_num(torch.round(((1.0 - v) * H - 1.0) * 0.5), nan=-4.0).clamp(-4, H + 4).long()
    offs = torch.arange(-win, win + 1)
    yi = (yc[:, None, None] + offs[None, :, None]).expand(n, 2 * win + 1, 2 * win + 1)
    xi = (xc[:, None, None] + offs[None, None, :]).expand(n, 2 * win + 1, 2 * win + 1)
    yf = 1.0 - (2.0 * yi.float() + 1.0) / H
    xf = 1.0 - (2.0 * xi.float() + 1.0) / W
    dx, dy = xf - u[:, None, None], yf - v[:, None, None]
    hit = (dx * dx + dy * dy < r2) & (d >= 0)[:, None, None] & (yi >= 0) & (yi < H) & (xi >= 0) & (xi < W)
    pix = (yi * W + xi)[hit]
    pid = torch.arange(n)[:, None, None].expand_as(hit)[hit]
    z = d[:, None, None].expand_as(hit)[hit]
    idx = torch.full((H * W,), -1, dtype=torch.int64)
    if pix.numel():
        # winner per pixel: smallest z, then smallest point index -> sort by (pixel, z, index), keep the first of each pixel
        order = np.lexsort((pid.numpy(), z.numpy(), pix.numpy()))
        pix_s, pid_s = pix.numpy()[order], pid.numpy()[order]
        first = np.ones(len(order), dtype=bool)
        first[1:] = pix_s[1:] != pix_s[:-1]
        idx[torch.from_numpy(pix_s[first])] = torch.from_numpy(pid_s[first])
    return idx.view(H, W)


FAST_RASTER = True


def surface_projection(points, cam, local_features, radius):
    """projection_model.py:127-157 for one sample: (N, C) features; sequential-assignment semantics for
    points owning several pixels (the LAST owned pixel in row-major order wins)."""
    C, H, W = local_features.shape
    idx = (rasterize_windowed if FAST_RASTER else rasterize_bruteforce)(points, cam, H, W, radius)
    out = torch.zeros(points.shape[0], C)
    visible = idx > -1
    pts = idx[visible]                       # row-major pixel order
    feats = local_features.permute(1, 2, 0)[visible]
    # later pixels overwrite earlier ones: keep, for every point, its LAST occurrence in row-major pixel order
    if pts.numel():
        last = torch.full((points.shape[0],), -1, dtype=torch.int64)
        last.scatter_reduce_(0, pts, torch.arange(pts.shape[0]), reduce="amax", include_self=True)
        sel = last >= 0
        out[sel] = feats[last[sel]]
    return out


def owner_pixels(points, cam, H, W, radius):
    """Per point: flat index of the last pixel it owns, or -1 (what bdm_rasterize_points returns)."""
    idx = rasterize_bruteforce(points, cam, H, W, radius).reshape(-1)
    own = torch.full((points.shape[0],), -1, dtype=torch.int64)
    pix = torch.nonzero(idx > -1).reshape(-1)
    for k in range(pix.shape[0]):  # ascending pixel order: later pixels overwrite earlier ones
        own[idx[pix[k]]] = pix[k]
    return own


def get_input_with_conditioning(x_t, cams, local_features, radius=0.0075):
    """projection_model.py:179-231: (B, N, 3 + C)."""
    proj = torch.stack([surface_projection(x_t[b], cams[b], local_features[b], radius) for b in range(x_t.shape[0])])
    return torch.cat([x_t, proj], dim=2)


# ---------------------------------------------------------------------------------------------
# loops
# ---------------------------------------------------------------------------------------------
TRACE = None  # optional callable(kind, t, x_after_step): progress / divergence curves in tools and tests


def _trace(kind, t, x):
    if TRACE is not None:
        TRACE(kind, t, x)


def interaction_sample(sd_pc2, x_t, cams, local_features, start_time, end_time, noises, prefix="point_cloud_model.model."):
    """model.py:216-291 with injected DDPM noise: noises[t] is the draw used at timestep t."""
    ddpm = RefDDPM()
    B = x_t.shape[0]
    for t in range(start_time - 1, end_time - 1, -1):
        x_in = get_input_with_conditioning(x_t, cams, local_features)
        eps = ref_net.point_cloud_model_forward(sd_pc2, x_in, torch.full((B,), t), prefix=prefix)
        x_t = ddpm.step(eps, t, x_t, noises.get(t) if t > 0 else None)
        _trace("pc2", t, x_t)
    return x_t


def pvd_prior(sd_pvd, points, start_time, end_time, noises, prefix="model.module."):
    """main_blending.py:175-183 + p_sample_loop: (B,N,3) in/out; noises[t] has shape (B,3,N)."""
    gd = RefPVDDiffusion()
    x = points.permute(0, 2, 1).float()
    B = x.shape[0]
    for t in range(start_time - 1, end_time - 1, -1):
        eps = ref_net.pvcnn_forward(sd_pvd, x, torch.full((B,), t), prefix=prefix)
        x = gd.step(eps, t, x, noises[t])
        _trace("pvd", t, x.permute(0, 2, 1))
    return x.permute(0, 2, 1)


def bdm_blending(sd_pc2, sd_pvd, x_init, cams, local_features, milestones, roll_step, recon_noise, branch_noise,
                 prior_noise, masks):
    """main_blending.py:232-344.  recon_noise[t] / branch_noise[t]: DDPM draws of the main chain / branch 1;
    prior_noise[t]: PVD draws; masks: list of (B,N) int64."""
    x = x_init - x_init.mean(dim=1, keepdim=True)
    times = len(milestones) - 1
    k = 0
    for i in range(times):
        if i == 0:
            x = interaction_sample(sd_pc2, x, cams, local_features, milestones[i], milestones[i + 1] - roll_step, recon_noise)
        elif i == times - 1:
            x = interaction_sample(sd_pc2, x, cams, local_features, milestones[i] - roll_step, milestones[i + 1], recon_noise)
        else:
            x = interaction_sample(sd_pc2, x, cams, local_features, milestones[i] - roll_step, milestones[i + 1], recon_noise)
            rec = interaction_sample(sd_pc2, x.clone(), cams, local_features, milestones[i + 1],
                                     milestones[i + 1] - roll_step, branch_noise)
            pri = pvd_prior(sd_pvd, x.clone(), milestones[i + 1], milestones[i + 1] - roll_step, prior_noise)
            m = masks[k].bool()[:, :, None]
            k += 1
            x = torch.where(m, pri, rec)
        _trace("segment", i, x)   # the cloud the next segment starts from (the last one = the result)
    return x


def nstep_fuse(sd_fuse, prior, recon, cams, local_features, t, z, prefix="fusion_model.model."):
    """model.py:510-570: centre both clouds, condition the recon cloud, PVCNN_fuse forward, one DDPM step on recon."""
    prior = prior - prior.mean(dim=1, keepdim=True)
    recon = recon - recon.mean(dim=1, keepdim=True)
    B = recon.shape[0]
    x_in = get_input_with_conditioning(recon, cams, local_features)
    eps = ref_net.pvcnn_fuse_forward(sd_fuse, x_in.transpose(1, 2), prior.transpose(1, 2), torch.full((B,), t), prefix=prefix,
                                     mode="fusion_nstep").transpose(1, 2)
    return RefDDPM().step(eps, t, recon, z if t > 0 else None)


def bdm_merging(sd_pc2, sd_pvd, sd_fuse, x_init, cams, local_features, fuse_local_features, milestones, roll_step, recon_noise,
                branch_noise, prior_noise, fuse_noise):
    """main_merging.py:415-520: branches run roll_step-1 steps, the last step of each window is the fused step."""
    x = x_init - x_init.mean(dim=1, keepdim=True)
    times = len(milestones) - 1
    for i in range(times):
        if i == 0:
            x = interaction_sample(sd_pc2, x, cams, local_features, milestones[i], milestones[i + 1] - roll_step, recon_noise)
        elif i == times - 1:
            x = interaction_sample(sd_pc2, x, cams, local_features, milestones[i] - roll_step, milestones[i + 1], recon_noise)
        else:
            x = interaction_sample(sd_pc2, x, cams, local_features, milestones[i] - roll_step, milestones[i + 1], recon_noise)
            rec = interaction_sample(sd_pc2, x.clone(), cams, local_features, milestones[i + 1],
                                     milestones[i + 1] - roll_step + 1, branch_noise)
            pri = pvd_prior(sd_pvd, x.clone(), milestones[i + 1], milestones[i + 1] - roll_step + 1, prior_noise)
            t = milestones[i + 1] - roll_step
            x = nstep_fuse(sd_fuse, pri, rec, cams, fuse_local_features, t, fuse_noise.get(t))
        _trace("segment", i, x)
    return x
