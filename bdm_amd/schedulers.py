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

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        t = int(timestep)
        c = self.step_coefficients(t)
        x = sample.contiguous()
        eps = model_output.contiguous()
        assert x.shape == eps.shape and x.dtype == torch.float32
        if self.streams is not None and self.noise_source is None:
            out = self.streams.ddpm_step(x, eps, dict(c, noise=t > 0), purpose=self.stream_purpose)
            return SimpleNamespace(prev_sample=out) if return_dict else (out,)
        noise = self._noise(x.shape, x.device, generator) if t > 0 else None
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


class _UnsupportedScheduler:
    """PNDM exists in the reference's schedulers_map (model.py:61) but no BDM recipe selects it; not built."""

    def __init__(self, name):
        self.name = name

    def __getattr__(self, item):
        raise NotImplementedError(f"{self.name} scheduler is not implemented on the MI355X path (DDPM and DDIM are)")


def make_schedulers_map(**scheduler_kwargs):
    return {"ddpm": DDPMScheduler(**scheduler_kwargs, clip_sample=False),
            "ddim": DDIMScheduler(**scheduler_kwargs, clip_sample=False), "pndm": _UnsupportedScheduler("pndm")}
