"""DDPM scheduler with the surface of `diffusers.DDPMScheduler` that the reference uses
(`requirements.txt:2` pins diffusers==0.21.0, NOT vendored in /root/reference and not installed
here, so its published algorithm is restated; call sites: experiments/model/model.py:58-62,
157-161,193,255,286,541,563).  Parity status: UNPINNED against diffusers itself (no reference
test or golden vector exists for it); the step is unit-tested against the closed-form DDPM
posterior in tests/test_sampler_host.py.

Restated behaviour (DDPMScheduler 0.21.0, prediction_type="epsilon", variance_type="fixed_small",
clip_sample=False, timestep_spacing="leading", steps_offset=0):
  betas = linspace(beta_start, beta_end, T, float32); alphas_cumprod = cumprod(1 - betas)
  set_timesteps(n): timesteps = (arange(n) * (T // n)).round()[::-1]      -> [999, ..., 0] for n = T
  step(eps, t, x):  prev_t = t - T // n
      abar_t = alphas_cumprod[t]; abar_prev = alphas_cumprod[prev_t] if prev_t >= 0 else 1
      alpha_t = abar_t / abar_prev; beta_t = 1 - alpha_t
      x0 = (x - sqrt(1 - abar_t) * eps) / sqrt(abar_t)
      mean = sqrt(abar_prev) * beta_t / (1 - abar_t) * x0 + sqrt(alpha_t) * (1 - abar_prev) / (1 - abar_t) * x
      if t > 0: mean += sqrt(clamp((1 - abar_prev) / (1 - abar_t) * beta_t, min=1e-20)) * randn(x.shape)
The elementwise arithmetic runs in bdm_ddpm_step (HIP); the scalars are float32 torch CPU ops in
the order above.  The timestep is taken as a Python int (no device->host sync per step; the
reference indexes a CPU table with a device scalar, i.e. one sync per step).
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib as L


class DDPMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon"):
        if trained_betas is not None:
            self.betas = torch.tensor(np.asarray(trained_betas), dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if variance_type != "fixed_small" or prediction_type != "epsilon":
            raise NotImplementedError("the BDM sampling path uses fixed_small variance and epsilon prediction")
        if clip_sample:
            raise NotImplementedError("the BDM models construct the scheduler with clip_sample=False (model.py:58)")
        self.num_train_timesteps = int(num_train_timesteps)
        self.config = SimpleNamespace(num_train_timesteps=self.num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      variance_type=variance_type, prediction_type=prediction_type)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())
        self.noise_source = None  # optional callable(shape, device) -> tensor: replay mode for parity tests
        self.streams = None       # optional rng.ShapeStreams: per-shape Philox noise generated inside the step kernel
        self.stream_purpose = 1   # rng.PC2 (the fusion model's scheduler uses rng.FUSE)

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = int(num_inference_steps)
        ratio = self.num_train_timesteps // self.num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def previous_timestep(self, t):
        n = self.num_inference_steps if self.num_inference_steps else self.num_train_timesteps
        return t - self.num_train_timesteps // n

    def step_coefficients(self, t):
        """float32 scalars of one step, computed op by op as DDPMScheduler.step does."""
        t = int(t)
        prev_t = self.previous_timestep(t)
        abar_t = self.alphas_cumprod[t]
        abar_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - abar_t
        beta_prod_prev = 1 - abar_prev
        cur_alpha = abar_t / abar_prev
        cur_beta = 1 - cur_alpha
        coef_x0 = (abar_prev ** 0.5 * cur_beta) / beta_prod_t
        coef_x = cur_alpha ** 0.5 * beta_prod_prev / beta_prod_t
        variance = torch.clamp((1 - abar_prev) / (1 - abar_t) * cur_beta, min=1e-20)
        return dict(sqrt_beta_prod=float(beta_prod_t ** 0.5), sqrt_alpha_prod=float(abar_t ** 0.5),
                    coef_x0=float(coef_x0), coef_x=float(coef_x), sigma=float(variance ** 0.5))

    def _noise(self, shape, device, generator):
        if self.noise_source is not None:
            return self.noise_source(tuple(shape), device)
        if self.streams is not None:
            return self.streams.normal(tuple(shape), self.stream_purpose)
        return torch.randn(shape, generator=generator, device=device, dtype=torch.float32)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, out=None):
        """out (an extension of the diffusers signature): a contiguous tensor the result is written to -- `sample` itself is allowed (the
        step is elementwise), which spares the reverse loops a copy launch per step."""
        t = int(timestep)
        c = self.step_coefficients(t)
        x = sample.contiguous()
        eps = model_output.contiguous()
        assert x.shape == eps.shape and x.dtype == torch.float32
        if out is not None and not (out.is_contiguous() and out.shape == x.shape and out.dtype == x.dtype and out.device == x.device):
            # (never a silent fresh tensor -- the in-place reverse loops discard the return value and would denoise the same x forever)
            raise ValueError(f"DDPMScheduler.step: `out` must be a contiguous {tuple(x.shape)} {x.dtype} tensor on {x.device}, got "
                             f"{tuple(out.shape)} {out.dtype} on {out.device}, contiguous={out.is_contiguous()}")
        if self.streams is not None and self.noise_source is None:
            out = self.streams.ddpm_step(x, eps, dict(c, noise=t > 0), purpose=self.stream_purpose, out=out)
            return SimpleNamespace(prev_sample=out) if return_dict else (out,)
        noise = self._noise(x.shape, x.device, generator) if t > 0 else None
        if out is None:
            out = torch.empty_like(x)
        L.check(L.lib().bdm_ddpm_step(L.c_ll(x.numel()), L.ptr(x), L.ptr(eps), L.ptr(noise), L.c_float(c["sqrt_beta_prod"]),
                                      L.c_float(c["sqrt_alpha_prod"]), L.c_float(c["coef_x0"]), L.c_float(c["coef_x"]),
                                      L.c_float(c["sigma"]), L.ptr(out), L.stream()), "ddpm_step")
        return SimpleNamespace(prev_sample=out) if return_dict else (out,)

    def coefficient_table(self, device):
        """(num_train_timesteps, 5) float32 on `device`: row t = the step's five scalars for the CURRENT set_timesteps
        grid, sigma = 0 at t = 0 (no noise).  Lets a captured hipGraph of one reverse step serve every timestep."""
        key = (self.num_inference_steps, str(device))
        hit = getattr(self, "_coef_table", None)
        if hit is None or hit[0] != key:
            rows = []
            for t in range(self.num_train_timesteps):
                c = self.step_coefficients(t)
                rows.append([c["sqrt_beta_prod"], c["sqrt_alpha_prod"], c["coef_x0"], c["coef_x"], c["sigma"] if t > 0 else 0.0])
            hit = (key, torch.tensor(rows, dtype=torch.float32).to(device))
            self._coef_table = hit
        return hit[1]

    def step_dev(self, model_output, coef_dev, sample, noise, out):
        """step() with the coefficients in device memory (coef_dev: 5 floats); graph-capturable, `out` may be `sample`."""
        eps = model_output.contiguous()
        assert sample.is_contiguous() and eps.shape == sample.shape == noise.shape
        L.check(L.lib().bdm_ddpm_step_dev(L.c_ll(sample.numel()), L.ptr(sample), L.ptr(eps), L.ptr(noise), L.ptr(coef_dev),
                                          L.ptr(out), L.stream()), "ddpm_step_dev")
        return out

    def add_noise(self, original_samples, noise, timesteps):
        """Training-side helper (model.py:99); plain tensor arithmetic, not on the sampling path."""
        ac = self.alphas_cumprod.to(original_samples.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise

    def __len__(self):
        return self.num_train_timesteps


class DDIMScheduler(DDPMScheduler):
    """diffusers 0.21.0 DDIMScheduler (epsilon prediction, clip_sample=False, set_alpha_to_one=True, leading spacing):
        prev_t = t - T // n;  abar_prev = alphas_cumprod[prev_t] if prev_t >= 0 else 1
        x0 = (x - sqrt(1 - abar_t) eps) / sqrt(abar_t)
        std = eta * sqrt((1 - abar_prev) / (1 - abar_t) * (1 - abar_t / abar_prev))
        x_prev = sqrt(abar_prev) x0 + sqrt(1 - abar_prev - std^2) eps  [+ std * randn  if eta > 0]
    Parity status: unpinned (diffusers absent), closed-form tested."""

    def step_coefficients(self, t, eta=0.0):
        t = int(t)
        prev_t = self.previous_timestep(t)
        abar_t = self.alphas_cumprod[t]
        abar_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - abar_t
        variance = (1 - abar_prev) / (1 - abar_t) * (1 - abar_t / abar_prev)
        std = eta * variance ** 0.5
        return dict(sqrt_beta_prod=float(beta_prod_t ** 0.5), sqrt_alpha_prod=float(abar_t ** 0.5),
                    coef_x0=float(abar_prev ** 0.5), coef_eps=float((1 - abar_prev - std ** 2) ** 0.5), sigma=float(std))

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, return_dict=True):
        c = self.step_coefficients(timestep, eta)
        x, eps = sample.contiguous(), model_output.contiguous()
        noise = self._noise(x.shape, x.device, generator) if eta > 0 else None
        out = torch.empty_like(x)
        L.check(L.lib().bdm_ddim_step(L.c_ll(x.numel()), L.ptr(x), L.ptr(eps), L.ptr(noise), L.c_float(c["sqrt_beta_prod"]),
                                      L.c_float(c["sqrt_alpha_prod"]), L.c_float(c["coef_x0"]), L.c_float(c["coef_eps"]),
                                      L.c_float(c["sigma"]), L.ptr(out), L.stream()), "ddim_step")
        return SimpleNamespace(prev_sample=out) if return_dict else (out,)


def _lincomb(terms, div=1.0):
    """(sum_i c_i * x_i) / div for up to four (coefficient, tensor) terms, one launch."""
    xs = [t.contiguous() for _, t in terms]
    cs = [float(c) for c, _ in terms] + [0.0] * (4 - len(terms))
    out = torch.empty_like(xs[0])
    ps = [L.ptr(x) for x in xs] + [L.ptr(None)] * (4 - len(xs))
    L.check(L.lib().bdm_lincomb(xs[0].numel(), len(xs), cs[0], ps[0], cs[1], ps[1], cs[2], ps[2], cs[3], ps[3], float(div),
                                L.ptr(out), L.stream()), "lincomb")
    return out


class PNDMScheduler(DDPMScheduler):
    """diffusers 0.21.0 PNDMScheduler as the reference constructs it (model/model.py:61: beta_start / beta_end /
    beta_schedule only -> skip_prk_steps=False, set_alpha_to_one=False, epsilon prediction, leading spacing, offset 0).
    Restated from the published algorithm (Liu et al., "Pseudo Numerical Methods for Diffusion Models on Manifolds",
    formulas (9), (12), (13)); unpinned against diffusers itself (absent), checked against oracle/ref_sampler.RefPNDM and
    closed-form properties.  No BDM recipe selects it; it completes the reference's schedulers_map.

      set_timesteps(n): ratio = T // n; base = arange(n) * ratio
          prk  = the last 4 base steps, each followed by its half step (+ ratio // 2), laid out as 12 Runge-Kutta stages
          plms = base[:-3] reversed;   timesteps = concat(prk, plms)             (n = 50 -> 59 network evaluations)
      step: 4-stage Runge-Kutta for the first 12 entries, then the linear multistep formulas of order 1..4 on the stored
      epsilons; every stage ends in   x_prev = sqrt(a_prev / a_t) x - (a_prev - a_t) e / (a_t sqrt(1 - a_prev) +
      sqrt(a_t (1 - a_t) a_prev))."""
    pndm_order = 4

    def __init__(self, *args, skip_prk_steps=False, set_alpha_to_one=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.skip_prk_steps = skip_prk_steps
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.config.skip_prk_steps, self.config.set_alpha_to_one = skip_prk_steps, set_alpha_to_one
        self._reset()

    def _reset(self):
        self.cur_model_output, self.counter, self.cur_sample, self.ets = None, 0, None, []

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = int(num_inference_steps)
        ratio = self.num_train_timesteps // self.num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64)
        self._timesteps = base
        if self.skip_prk_steps:
            self.prk_timesteps = np.array([], dtype=np.int64)
            self.plms_timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        else:
            prk = np.array(base[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, ratio // 2]), self.pndm_order)
            self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = base[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self._reset()

    def _prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[int(timestep)]
        a_prev = self.alphas_cumprod[int(prev_timestep)] if prev_timestep >= 0 else self.final_alpha_cumprod
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * (1 - a_prev) ** 0.5 + (a_t * (1 - a_t) * a_prev) ** 0.5
        return _lincomb([(float(sample_coeff), sample), (-float((a_prev - a_t) / denom), model_output)])

    def step(self, model_output, timestep, sample, return_dict=True, **_ignored):
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        t, ratio = int(timestep), self.num_train_timesteps // self.num_inference_steps
        e = model_output.contiguous()
        if self.counter < len(self.prk_timesteps) and not self.skip_prk_steps:   # ---- Runge-Kutta stage
            prev_t = t - (0 if self.counter % 2 else ratio // 2)
            t = int(self.prk_timesteps[self.counter // 4 * 4])
            stage = self.counter % 4
            if stage == 0:
                self.cur_model_output = _lincomb([(1 / 6, e)])
                self.ets.append(e)
                self.cur_sample = sample
            elif stage in (1, 2):
                self.cur_model_output = _lincomb([(1.0, self.cur_model_output), (1 / 3, e)])
            else:
                e = _lincomb([(1.0, self.cur_model_output), (1 / 6, e)])
                self.cur_model_output = None
            cur = self.cur_sample if self.cur_sample is not None else sample
            out = self._prev_sample(cur, t, prev_t, e)
        else:                                                                     # ---- linear multistep
            prev_t = t - ratio
            if self.counter != 1:
                self.ets = self.ets[-3:]
                self.ets.append(e)
            else:
                prev_t, t = t, t + ratio
            if len(self.ets) == 1 and self.counter == 0:
                self.cur_sample = sample
            elif len(self.ets) == 1 and self.counter == 1:
                e = _lincomb([(1.0, e), (1.0, self.ets[-1])], div=2.0)
                sample, self.cur_sample = self.cur_sample, None
            elif len(self.ets) == 2:
                e = _lincomb([(3.0, self.ets[-1]), (-1.0, self.ets[-2])], div=2.0)
            elif len(self.ets) == 3:
                e = _lincomb([(23.0, self.ets[-1]), (-16.0, self.ets[-2]), (5.0, self.ets[-3])], div=12.0)
            else:
                e = _lincomb([(55.0, self.ets[-1]), (-59.0, self.ets[-2]), (37.0, self.ets[-3]), (-9.0, self.ets[-4])], div=24.0)
            out = self._prev_sample(sample, t, prev_t, e)
        self.counter += 1
        return SimpleNamespace(prev_sample=out) if return_dict else (out,)


def make_schedulers_map(**scheduler_kwargs):
    return {"ddpm": DDPMScheduler(**scheduler_kwargs, clip_sample=False),
            "ddim": DDIMScheduler(**scheduler_kwargs, clip_sample=False),
            "pndm": PNDMScheduler(**scheduler_kwargs, clip_sample=False)}
