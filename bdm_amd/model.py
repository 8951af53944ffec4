"""Diffusion-model API of the reference on the HIP path (experiments/model/{model,projection_model,
point_cloud_model,__init__}.py): same class names, constructor arguments, attributes
(`schedulers_map`, `scheduler`, `point_cloud_model`, `feature_model`, `fusion_model`), methods
(`forward`, `forward_sample`, `interaction_sample`, `nstep_fuse`, `get_input_with_conditioning`)
and state-dict keys (`point_cloud_model.model.*`, `feature_model.model.*`, `fusion_model.model.*`),
so main_blending.py / main_merging.py drive it unchanged.  Sampling only.
"""
from typing import Optional

import os

import numpy as np

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib as L
from . import ops
from .cameras import PerspectiveCameras, Pointclouds, join_cameras
from .feature_model import FeatureModel
from .pvcnn import PVCNN2_PC2, PVCNN_fuse
from .schedulers import DDIMScheduler, DDPMScheduler, make_schedulers_map


def compute_distance_transform(mask: Tensor):
    """model_utils.py:13-21: cv2.distanceTransform(1 - m, DIST_L2, DIST_MASK_3) / (image_size / 2), clipped to [0, 1]:
    distance of every background pixel to the nearest foreground pixel.  cv2 is absent here; its 3x3 DIST_L2 transform is
    the two-pass chamfer transform with weights a = 0.955 (edge neighbours), b = 1.3693 (diagonal neighbours) in 16-bit
    fixed point (OpenCV documentation of distanceTransform; "unpinned": restated, checked against the exact Euclidean
    transform within the chamfer metric's known error in tests/test_sampler_host.py).  Host code, once per image batch."""
    image_size = mask.shape[-1]
    HV, DIAG, BIG = int(round(0.955 * 65536)), int(round(1.3693 * 65536)), (2 ** 31 - 1) >> 2
    outs = []
    for m in mask.squeeze(1).detach().cpu().numpy().astype(np.uint8):
        src = 1 - m                                   # zero where the mask is set
        H, W = src.shape
        d = np.full((H + 2, W + 2), BIG, dtype=np.int64)
        d[1:-1, 1:-1] = np.where(src == 0, 0, BIG)
        for i in range(1, H + 1):                     # forward pass: up-left, up, up-right, left
            for j in range(1, W + 1):
                if d[i, j]:
                    d[i, j] = min(d[i, j], d[i - 1, j - 1] + DIAG, d[i - 1, j] + HV, d[i - 1, j + 1] + DIAG, d[i, j - 1] + HV)
        for i in range(H, 0, -1):                     # backward pass: right, down-right, down, down-left
            for j in range(W, 0, -1):
                if d[i, j]:
                    d[i, j] = min(d[i, j], d[i, j + 1] + HV, d[i + 1, j + 1] + DIAG, d[i + 1, j] + HV, d[i + 1, j - 1] + DIAG)
        outs.append(torch.from_numpy((d[1:-1, 1:-1].astype(np.float64) / 65536.0).astype(np.float32) / (image_size / 2)))
    return torch.stack(outs).unsqueeze(1).clip(0, 1).to(mask.device)


class _DeviceMixin:
    @property
    def device(self):
        return next(self.parameters()).device


class PointCloudModel(_DeviceMixin, nn.Module):
    """point_cloud_model.py:14-65 (model_type 'pvcnn')."""

    def __init__(self, model_type="pvcnn", in_channels=3, out_channels=3, embed_dim=64, dropout=0.1,
                 width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        if model_type != "pvcnn":
            raise NotImplementedError("only the default 'pvcnn' denoiser (config/structured.py:110) is on the HIP path")
        self.model_type = model_type
        self.model = PVCNN2_PC2(embed_dim=embed_dim, num_classes=out_channels, extra_feature_channels=in_channels - 3,
                                dropout=dropout, width_multiplier=width_multiplier,
                                voxel_resolution_multiplier=voxel_resolution_multiplier)
        self.model.classifier[-1].bias.data.normal_(0, 1e-6)
        self.model.classifier[-1].weight.data.normal_(0, 1e-6)

    @torch.no_grad()
    def forward(self, inputs: Tensor, t: Tensor) -> Tensor:
        """(B, N, in_channels) -> (B, N, out_channels)."""
        cond = getattr(inputs, "_bdm_cond", None)  # projection conditioning in factored form (ops.Conditioning)
        if cond is not None and not ops.is_transposed_view_of(inputs, cond.x_cf):
            # the handle will not travel on (transpose12 is going to COPY): complete the lazily built tensor BEFORE it is copied
            cond.ensure_features()
        x = ops.transpose12(inputs)
        if cond is not None and cond.x_cf.data_ptr() == x.data_ptr():
            x._bdm_cond = cond
        elif cond is not None:
            cond.ensure_features()
            x = ops.transpose12(inputs)   # (unreachable unless transpose12's view test and the one above disagree: redo the copy)
        return ops.transpose12(self.model(x, t))


class PC2_PVDFusionModel(_DeviceMixin, nn.Module):
    """point_cloud_model.py:68-100."""

    def __init__(self, pvd_model, pc2_model, in_channels=3, out_channels=3, embed_dim=64, dropout=0.1,
                 width_multiplier=1, voxel_resolution_multiplier=1):
        super().__init__()
        self.model = PVCNN_fuse(pvd_model=pvd_model, pc2_model=pc2_model, embed_dim=embed_dim, num_classes=out_channels,
                                extra_feature_channels=in_channels - 3, dropout=dropout,
                                width_multiplier=width_multiplier, voxel_resolution_multiplier=voxel_resolution_multiplier)

    @torch.no_grad()
    def forward(self, input_with_condition, pred_from_prior, t, mode="fusion_nstep"):
        return ops.transpose12(self.model(ops.transpose12(input_with_condition), ops.transpose12(pred_from_prior), t, mode))


class PointCloudProjectionModel(_DeviceMixin, nn.Module):
    """projection_model.py:19-235.  The image encoder is hoisted out of the per-step loop: the pixel-major
    conditioning image is cached per `image_rgb` tensor and only the x_t-dependent part (rasterise the current
    points, copy each owned pixel's feature vector) runs at every step, in HIP."""

    def __init__(self, image_size: int, image_feature_model: str, use_local_colors=True, use_local_features=True,
                 use_global_features=False, use_mask=True, use_distance_transform=True, predict_shape=True,
                 predict_color=False, process_color=False, image_color_channels=3, color_channels=3, colors_mean=0.5,
                 colors_std=0.5, scale_factor=1.0, raster_point_radius=0.0075, raster_points_per_pixel=1, bin_size=0):
        super().__init__()
        self.image_size = image_size
        self.scale_factor = scale_factor
        self.use_local_colors, self.use_local_features = use_local_colors, use_local_features
        self.use_global_features, self.use_mask = use_global_features, use_mask
        self.use_distance_transform = use_distance_transform
        self.predict_shape, self.predict_color, self.process_color = predict_shape, predict_color, process_color
        self.image_color_channels, self.color_channels = image_color_channels, color_channels
        self.colors_mean, self.colors_std = colors_mean, colors_std
        if use_global_features or predict_color or process_color:
            raise NotImplementedError("global features / colour prediction are not used by the BDM recipes (config/structured.py:80-90)")
        if use_distance_transform and not use_mask:
            raise ValueError("No mask for distance transform?")  # projection_model.py:119-120
        if raster_points_per_pixel != 1:
            raise NotImplementedError("one point per pixel (projection_model.py:41)")
        self.use_local_conditioning = use_local_colors or use_local_features or use_mask
        self.use_global_conditioning = False
        self.feature_model = FeatureModel(image_size, image_feature_model)
        self.in_channels = 3 + (image_color_channels if use_local_colors else 0) + \
            (self.feature_model.feature_dim if use_local_features else 0) + \
            ((2 if use_distance_transform else 1) if use_mask else 0)  # projection_model.py:67-77
        self.out_channels = 3
        self.raster_point_radius = raster_point_radius
        self._cond_cache = None

    def normalize(self, x):
        return (x - self.colors_mean) / self.colors_std

    def denormalize(self, x, clamp=True):
        x = x * self.colors_std + self.colors_mean
        return torch.clamp(x, 0, 1) if clamp else x

    def get_local_conditioning(self, image_rgb, mask=None):
        """projection_model.py:110-125 -> (B, D_cond, H, W)."""
        parts = []
        if self.use_local_colors:
            parts.append(self.normalize(image_rgb))
        if self.use_local_features:
            parts.append(self.feature_model(image_rgb))
        parts.extend(self._mask_channels(image_rgb, mask))
        return torch.cat(parts, dim=1)

    def _mask_channels(self, image_rgb, mask):
        """[mask] (+ [distance transform]) as (B, 1, H, W) float tensors (projection_model.py:116-123)."""
        if not self.use_mask:
            return []
        if mask is None:
            raise ValueError("use_mask=True needs batch.fg_probability")
        out = [mask.float().to(image_rgb.device)]
        if self.use_distance_transform:
            m = (mask > 0.5) if mask.is_floating_point() else mask
            out.append(compute_distance_transform(m).to(image_rgb.device))
        return out

    def conditioning_image(self, image_rgb, mask=None):
        """Pixel-major (B, H*W, D_cond) conditioning image, computed once per image batch (hoisted)."""
        # The entry holds the image tensor itself and is matched by identity (+ in-place version): a freed batch's
        # address can be handed to the next batch by the caching allocator, so a data_ptr key alone would alias.
        hit = self._cond_cache
        if hit is None or hit[0] is not image_rgb or hit[1] != image_rgb._version or hit[4] is not mask:
            assert self.use_local_colors and self.use_local_features
            H, W = image_rgb.shape[-2:]
            img = self.feature_model.conditioning_image(image_rgb, self.colors_mean, self.colors_std)   # (B, H*W, 3 + D)
            extra = self._mask_channels(image_rgb, mask)
            if extra:  # once per image batch (hoisted): the mask / distance-transform channels join the pixel-major image
                img = torch.cat([img] + [e.reshape(e.shape[0], H * W, 1) for e in extra], dim=2).contiguous()
            # A same-shaped image batch rewrites the conditioning image and its hoisted maps IN PLACE: a recorded step (launch
            # tape) that holds their addresses stays valid for the new images -- no re-recording per trajectory / per batch
            maps, prev = {}, getattr(self, "_cond_static", None)
            if prev is not None and prev[0].shape == img.shape and prev[0].device == img.device and prev[0].dtype == img.dtype:
                prev[0].copy_(img)
                img, maps = prev
                ops.refresh_conditioning_maps(img, maps)
            self._cond_static = (img, maps)
            hit = (image_rgb, image_rgb._version, img, (H, W), mask, maps)  # maps: hoisted maps of this image batch (ops.Conditioning)
            self._cond_cache = hit
        return hit[2], hit[3]

    def surface_projection_indices(self, points, camera, hw):
        """Per-point owning pixel (or -1): rasterisation part of projection_model.py:127-157."""
        B, N, _ = points.shape
        H, W = hw
        # packed camera parameters live on the device for the lifetime of the batch: a per-step pageable host->device
        # copy would block the host until the stream drains and serialise CPU launch work with the GPU
        key = (id(camera), str(points.device), float(self.scale_factor))
        hit = getattr(self, "_cam_cache", None)
        if hit is None or hit[0] != key:
            cam = join_cameras(camera).clone()
            cam.T = cam.T * self.scale_factor
            hit = (key, cam.packed().to(points.device), camera)  # keeps `camera` alive so its id stays unique
            self._cam_cache = hit
        cams = hit[1]
        assert cams.shape[0] == B
        points = points.contiguous()  # bound to a local: raw pointers must not outlive their tensor
        pix = torch.empty(B, N, dtype=torch.int32, device=points.device)
        ws = ops.workspace(L.lib().bdm_rasterize_workspace_bytes(B, H, W), points.device, "raster")
        L.check(L.lib().bdm_rasterize_points(B, N, H, W, L.c_float(self.raster_point_radius), L.ptr(points),
                                             L.ptr(cams), L.ptr(pix), L.ptr(ws), L.stream()), "rasterize_points")
        return pix

    def point_cloud_to_tensor(self, pc, normalize=False, scale=False):
        if isinstance(pc, torch.Tensor):
            return pc
        return pc.points_padded() * (self.scale_factor if scale else 1)

    def tensor_to_point_cloud(self, x, denormalize=False, unscale=False):
        assert x.shape[2] == 3
        return Pointclouds(points=x[:, :, :3] / (self.scale_factor if unscale else 1))

    @torch.no_grad()
    def get_input_with_conditioning(self, x_t, camera, image_rgb, mask, t, lazy=False):
        """projection_model.py:179-231 -> (B, N, in_channels).  lazy (the reverse loops only): the feature channels of the result may be
        left unwritten -- the tensor then carries a handle (ops.Conditioning, features_ready False) and the denoiser either reads the
        hoisted maps instead or completes it; nothing else may read it."""
        B, N = x_t.shape[:2]
        x_t = x_t.contiguous()
        if not self.use_local_conditioning:
            return x_t
        feat, hw = self.conditioning_image(image_rgb, mask)
        if hw != tuple(image_rgb.shape[-2:]):
            raise ValueError(f"{hw=} and {image_rgb.shape=}")
        early = None
        if CHANNEL_FIRST_CONDITIONING and ops.HOIST_CONDITIONING and x_t.shape[2] == 3 and x_t.is_cuda:
            from . import pvcnn
            early = pvcnn.early_first_sampler(getattr(getattr(self, "point_cloud_model", None), "model", None), x_t)
        pix = self.surface_projection_indices(x_t[:, :, :3], camera, hw)
        C = feat.shape[2]
        if CHANNEL_FIRST_CONDITIONING:
            # written channel-first and returned as its (B, N, 3 + C) transposed VIEW: the reference's shape and values, and the
            # denoiser's `inputs.transpose(1, 2)` (point_cloud_model.py:65) becomes free (ops.transpose12 sees the view)
            out = torch.empty(B, 3 + C, N, dtype=torch.float32, device=x_t.device)
            lazy = bool(lazy and LAZY_CONDITIONING and ops.HOIST_CONDITIONING and x_t.shape[2] == 3 and x_t.is_cuda)
            if lazy:
                L.check(L.lib().bdm_condition_xyz_cf(B, N, C, L.ptr(x_t), L.ptr(out), L.stream()), "condition_xyz_cf")
            else:
                L.check(L.lib().bdm_condition_gather_cf(B, N, C, hw[0] * hw[1], L.ptr(x_t), L.ptr(feat), L.ptr(pix), L.ptr(out),
                                                        L.stream()), "condition_gather_cf")
            res = out.transpose(1, 2)
            if ops.HOIST_CONDITIONING and x_t.shape[2] == 3:
                res._bdm_cond = ops.Conditioning(feat, hw, pix, x_t, out, self._cond_cache[5])
                res._bdm_cond.early = early
                res._bdm_cond.features_ready = not lazy
            return res
        out = torch.empty(B, N, 3 + C, dtype=torch.float32, device=x_t.device)
        L.check(L.lib().bdm_condition_gather(B, N, C, hw[0] * hw[1], L.ptr(x_t), L.ptr(feat), L.ptr(pix), L.ptr(out),
                                             L.stream()), "condition_gather")
        return out


LAZY_CONDITIONING = os.environ.get("BDM_LAZY_CONDITIONING", "1") == "1"  # reverse loops: feature rows of x_in written only if a layer reads them
CHANNEL_FIRST_CONDITIONING = True  # the conditioning gather writes the denoiser's channel-first input directly (tests flip the attribute)


def _timestep_list(scheduler, num_inference_steps, start_time, end_time):
    scheduler.set_timesteps(num_inference_steps)
    ts = scheduler.timesteps[num_inference_steps - start_time: num_inference_steps - end_time]
    return [int(v) for v in ts]


# BDM_GRAPH=1 captures the PC^2 reverse step into a hipGraph and replays it per timestep.  Opt-in: on ROCm 7.2 the replay
# of the ~300-node step costs the host as much as enqueueing the kernels one by one (measured 4.5 vs 4.8 ms per step at
# B=1, identical GPU time at B=16: tools/time_loop.py), so the eager loop stays the default.
GRAPH_STEPS = os.environ.get("BDM_GRAPH", "0") == "1"
GRAPH_MIN_STEPS = 8  # shorter segments do not amortise the capture
# BDM_TAPE: "auto" (default) replays a recorded launch tape (tape.py) for host-bound problem sizes, "1" always, "0" never.
TAPE_STEPS = os.environ.get("BDM_TAPE", "auto")
TAPE_MAX_POINTS = 1 << 20  # B * N up to which "auto" takes the tape: every BASELINE.json configuration (C5: 32 x 16384).  Recorded
# inside a private memory pool the replayed step costs the GPU what the eager step costs, and the host 1 ms instead of 5


class ConditionalPointCloudDiffusionModel(PointCloudProjectionModel):
    """model.py:23-318."""

    def __init__(self, beta_start: float, beta_end: float, beta_schedule: str, point_cloud_model: str,
                 point_cloud_model_embed_dim: int, **kwargs):
        super().__init__(**kwargs)
        if not self.predict_shape:
            raise NotImplementedError("Must predict shape if performing diffusion.")
        if beta_schedule == "custom":
            raise NotImplementedError("custom beta schedule is not used by the BDM recipes")
        self.schedulers_map = make_schedulers_map(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule)
        self.scheduler = self.schedulers_map["ddpm"]
        self.point_cloud_model = PointCloudModel(model_type=point_cloud_model, embed_dim=point_cloud_model_embed_dim,
                                                 in_channels=self.in_channels, out_channels=self.out_channels)

    def _weights_signature(self):
        """Changes whenever a denoiser parameter is rewritten or replaced: a recorded step (graph or tape) holds the addresses
        of the weight packs derived from them, so it must not outlive the weights it was recorded with."""
        return hash(tuple((p.data_ptr(), p._version) for p in self.point_cloud_model.parameters()))

    _step_kwargs = {}  # e.g. {"eta": 0.0} for DDIM (the reference forwards eta only to schedulers that accept it)

    def _denoise_loop(self, x_t, camera, image_rgb, mask, scheduler, timesteps, generator=None):
        long_enough = x_t.is_cuda and len(timesteps) >= GRAPH_MIN_STEPS
        if (GRAPH_STEPS and long_enough and type(scheduler) is DDPMScheduler and not self._step_kwargs
                and getattr(scheduler, "streams", None) is None):
            return self._denoise_loop_graph(x_t, camera, image_rgb, mask, scheduler, timesteps, generator)
        # the tape holds the denoiser forward only; schedulers that keep earlier model outputs (PNDM) would see them overwritten
        if (long_enough and type(scheduler) in (DDPMScheduler, DDIMScheduler)
                and (TAPE_STEPS == "1" or (TAPE_STEPS == "auto" and x_t.shape[0] * x_t.shape[1] <= TAPE_MAX_POINTS))):
            return self._denoise_loop_tape(x_t, camera, image_rgb, mask, scheduler, timesteps, generator)
        B = x_t.shape[0]
        for t in timesteps:
            tt = torch.full((B,), t, dtype=torch.int64, device=x_t.device)
            x_in = self.get_input_with_conditioning(x_t, camera=camera, image_rgb=image_rgb, mask=mask, t=tt, lazy=True)
            noise_pred = self.point_cloud_model(x_in, tt)
            x_t = scheduler.step(noise_pred, t, x_t, generator=generator, **self._step_kwargs).prev_sample
        return x_t

    # ---- hipGraph form of the reverse loop -------------------------------------------------------------------
    # One reverse step (rasterise + feature gather, the ~300 launches of the denoiser incl. its side-stream sampler
    # chain, the scheduler arithmetic) is captured ONCE on static buffers and replayed per timestep: the timestep
    # tensor, the five scheduler scalars and the step's Gaussian noise are refreshed in device memory between replays
    # (three tiny eager ops), so the host enqueues one graph instead of ~300 kernels per step.  Values are identical
    # to the eager loop (same kernels, same arguments, same RNG stream: the noise is drawn eagerly, in order).
    def _step_graph(self, x_t, camera, image_rgb, mask, scheduler):
        feat, _ = self.conditioning_image(image_rgb, mask)
        key = (tuple(x_t.shape), str(x_t.device), id(camera), id(scheduler), scheduler.num_inference_steps, self._weights_signature(),
               ops.saturation_epoch())
        g = getattr(self, "_graph_cache", None)
        if g is not None and g["key"] == key and g["feat"] is feat and g["image"] is image_rgb:
            return g
        dev, B = x_t.device, x_t.shape[0]
        g = {"key": key, "feat": feat, "image": image_rgb, "camera": camera,
             "x": torch.empty_like(x_t, memory_format=torch.contiguous_format), "noise": torch.zeros_like(x_t),
             "t": torch.zeros(B, dtype=torch.int64, device=dev), "coef": torch.ones(5, dtype=torch.float32, device=dev)}
        g["x"].copy_(x_t)

        def step():
            x_in = self.get_input_with_conditioning(g["x"], camera=camera, image_rgb=image_rgb, mask=mask, t=g["t"], lazy=True)
            eps = self.point_cloud_model(x_in, g["t"])
            scheduler.step_dev(eps, g["coef"], g["x"], g["noise"], out=g["x"])

        cur = torch.cuda.current_stream()
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):  # allocator, workspaces, weight packs, kernel attributes: everything lazy happens here
            step()
        cur.wait_stream(warm)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with ops.static_step(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
            step()
        g["graph"] = graph
        self._graph_cache = g
        return g

    def _denoise_loop_graph(self, x_t, camera, image_rgb, mask, scheduler, timesteps, generator=None):
        g = self._step_graph(x_t, camera, image_rgb, mask, scheduler)
        table = scheduler.coefficient_table(x_t.device)
        g["x"].copy_(x_t)
        probe = int(getattr(self, "eager_probe_every", 0))  # bench.py: every k-th step runs eagerly so that single
        for i, t in enumerate(timesteps):                     # kernels can be timed with HIP events inside the loop
            if probe and i % probe == probe - 1:
                tt = torch.full((x_t.shape[0],), t, dtype=torch.int64, device=x_t.device)
                x_in = self.get_input_with_conditioning(g["x"], camera=camera, image_rgb=image_rgb, mask=mask, t=tt, lazy=True)
                g["x"].copy_(scheduler.step(self.point_cloud_model(x_in, tt), t, g["x"], generator=generator).prev_sample)
                continue
            g["t"].fill_(t)
            g["coef"].copy_(table[t])
            if t > 0:
                g["noise"].copy_(scheduler._noise(g["x"].shape, x_t.device, generator))
            g["graph"].replay()
        return g["x"].clone()

    # ---- launch-tape form of the reverse loop (tape.py) --------------------------------------------------------
    # The conditioning + denoiser forward on static buffers is RECORDED while it runs eagerly (second step of the loop: the
    # first one leaves every lazy cache -- weight packs, workspaces, kernel attributes -- behind it) and later steps
    # replay the flat list of C-ABI calls: ~11 us of host time per launch instead of 15-20.  The scheduler step stays
    # eager, so DDPM and DDIM, generator / noise_source draws and the per-shape Philox streams all work unchanged.
    # Default for every configuration up to B * N = TAPE_MAX_POINTS: recorded inside a private memory pool (tape.py) the replay
    # costs the GPU what the eager step costs, and the host one library call per step.
    def _denoise_loop_tape(self, x_t, camera, image_rgb, mask, scheduler, timesteps, generator=None):
        from . import tape as T
        feat, _ = self.conditioning_image(image_rgb, mask)
        dev, B = x_t.device, x_t.shape[0]
        key = (tuple(x_t.shape), str(dev), id(camera), torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()),
               self._weights_signature(), ops.saturation_epoch())
        g = getattr(self, "_tape_cache", None)
        if g is None or g["key"] != key or g["feat"] is not feat:  # (feat is a static buffer: conditioning_image)
            g = {"key": key, "feat": feat, "image": image_rgb, "camera": camera, "tape": None, "warm": False, "off": None, "eps": None,
                 "x": torch.empty_like(x_t, memory_format=torch.contiguous_format),
                 "t": torch.zeros(B, dtype=torch.int64, device=dev)}
            self._tape_cache = g

        def denoise():  # conditioning + denoiser on the static buffers: everything between two scheduler steps
            x_in = self.get_input_with_conditioning(g["x"], camera=camera, image_rgb=image_rgb, mask=mask, t=g["t"], lazy=True)
            return self.point_cloud_model(x_in, g["t"])

        g["x"].copy_(x_t)
        probe = int(getattr(self, "eager_probe_every", 0))  # bench.py: every k-th step runs eagerly, through the kernel-class
        from . import profiling                               # profiler (HIP events around single launches inside the timed loop)
        for t in timesteps:
            g["t"].fill_(t)
            g["steps"] = i = g.get("steps", -1) + 1  # counted across calls: the samplers run the loop in segments of 32 ... 744 steps
            if g["tape"] is not None and probe and i % probe == probe - 1:
                profiling.PROBE_WEIGHT[0] = probe
                try:
                    eps = denoise()
                finally:
                    profiling.PROBE_WEIGHT[0] = 1
            elif g["tape"] is not None:
                g["tape"].replay()
                eps = g["eps"]
            elif g["warm"] and g["off"] is None:
                with ops.static_step(), T.record() as tp:
                    eps = denoise()
                if tp.broken:
                    g["off"] = tp.broken  # stay eager (on the static buffers) and say why: model._tape_cache["off"]
                else:
                    g["tape"], g["eps"] = tp, eps
            else:
                eps = denoise()
                g["warm"] = True
            # the scheduler step stays eager (one launch): its scalars, its noise draw and the per-shape Philox streams change per step
            if type(scheduler) is DDPMScheduler and not self._step_kwargs:
                scheduler.step(eps, t, g["x"], generator=generator, out=g["x"])   # in place: the step is elementwise (no copy launch)
            else:
                g["x"].copy_(scheduler.step(eps, t, g["x"], generator=generator, **self._step_kwargs).prev_sample)
        return g["x"].clone()

    @torch.no_grad()
    def forward_sample(self, num_points: int, camera, image_rgb: Optional[Tensor], mask: Optional[Tensor],
                       scheduler: Optional[str] = "ddpm", num_inference_steps: Optional[int] = 1000, eta=0.0,
                       return_sample_every_n_steps: int = -1, disable_tqdm: bool = False):
        """model.py:123-214 (vanilla PC^2 sampling)."""
        scheduler = self.scheduler if scheduler is None else self.schedulers_map[scheduler]
        B = 1 if image_rgb is None else image_rgb.shape[0]
        device = self.device if image_rgb is None else image_rgb.device
        x_t = torch.randn(B, num_points, 3, device=device)
        scheduler.set_timesteps(num_inference_steps)
        ts = [int(v) for v in scheduler.timesteps]  # ALL entries (model.py:182): PNDM lists more than num_inference_steps
        all_outputs = []
        if return_sample_every_n_steps > 0:
            for i, t in enumerate(ts):
                x_t = self._denoise_loop(x_t, camera, image_rgb, mask, scheduler, [t])
                if i % return_sample_every_n_steps == 0 or i == len(ts) - 1:
                    all_outputs.append(x_t)
            return self.tensor_to_point_cloud(x_t, denormalize=True, unscale=True), \
                [self.tensor_to_point_cloud(o, denormalize=True, unscale=True) for o in all_outputs]
        x_t = self._denoise_loop(x_t, camera, image_rgb, mask, scheduler, ts)
        ops.poll_h2_saturation()  # fp16x3 accuracy guard: one host check per trajectory
        return self.tensor_to_point_cloud(x_t, denormalize=True, unscale=True)

    @torch.no_grad()
    def interaction_sample(self, point_cloud: Tensor, camera, image_rgb: Tensor, mask: Optional[Tensor],
                           scheduler: Optional[str] = "ddpm", num_inference_steps: Optional[int] = 1000, eta=0.0,
                           return_sample_every_n_steps: int = -1, disable_tqdm: bool = False,
                           start_time: int = 1000, end_time: int = 0):
        """model.py:216-291: PC^2 steps t = start_time-1 ... end_time on `point_cloud` (B, N, 3)."""
        scheduler = self.scheduler if scheduler is None else self.schedulers_map[scheduler]
        ts = _timestep_list(scheduler, num_inference_steps, start_time, end_time)
        return self._denoise_loop(point_cloud, camera, image_rgb, mask, scheduler, ts)

    def forward(self, batch, mode: str = "train", **kwargs):
        if isinstance(batch, dict):
            from .data import FrameData
            batch = FrameData(**batch)
        if mode == "sample":
            pc = batch.sequence_point_cloud
            num_points = kwargs.pop("num_points", pc.points_padded().shape[1] if isinstance(pc, Pointclouds) else pc.shape[1])
            return self.forward_sample(num_points=num_points, camera=batch.camera, image_rgb=batch.image_rgb,
                                       mask=batch.fg_probability, **kwargs)
        raise NotImplementedError("training is out of scope for the MI355X sampling path")


class PointCloudFusionModel(PointCloudProjectionModel):
    """model.py:320-600 (sampling half: nstep_fuse)."""

    def __init__(self, pvd_model, pc2_model, beta_start: float, beta_end: float, beta_schedule: str,
                 point_cloud_model: str, point_cloud_model_embed_dim: int, **kwargs):
        super().__init__(**kwargs)
        self.p_forget = 0.2
        self.schedulers_map = make_schedulers_map(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule)
        self.scheduler = self.schedulers_map["ddpm"]
        self.fusion_model = PC2_PVDFusionModel(pc2_model=pc2_model, pvd_model=pvd_model,
                                               embed_dim=point_cloud_model_embed_dim, in_channels=self.in_channels,
                                               out_channels=self.out_channels)

    @torch.no_grad()
    def nstep_fuse(self, pred_from_prior: Tensor, pred_from_recon: Tensor, camera, image_rgb: Tensor,
                   mask: Optional[Tensor], scheduler: Optional[str] = "ddpm", num_inference_steps: Optional[int] = 1000,
                   eta=0.0, return_sample_every_n_steps: int = -1, disable_tqdm: bool = False, timestep: int = 0):
        """model.py:510-570: one fused denoising step at `timestep` on the recon cloud."""
        scheduler = self.scheduler if scheduler is None else self.schedulers_map[scheduler]
        B, N = pred_from_recon.shape[:2]
        # in-place centring of BOTH inputs, as the reference does (model.py:530-531)
        for x in (pred_from_prior, pred_from_recon):
            assert x.is_contiguous()
            L.check(L.lib().bdm_center_points(B, N, L.ptr(x), L.stream()), "center_points")
        scheduler.set_timesteps(num_inference_steps)
        t = int(timestep)
        tt = torch.full((B,), t, dtype=torch.int64, device=pred_from_recon.device)
        x_in = self.get_input_with_conditioning(pred_from_recon, camera=camera, image_rgb=image_rgb, mask=mask, t=tt)
        noise_pred = self.fusion_model(x_in, pred_from_prior, tt, mode="fusion_nstep")
        return scheduler.step(noise_pred, t, pred_from_recon).prev_sample


def get_model(cfg):
    """model/__init__.py:7-11."""
    return ConditionalPointCloudDiffusionModel(**cfg.model.as_kwargs())


def get_fusion_model(cfg, pvd_model, pc2_model):
    """model/__init__.py:21-36."""
    model = PointCloudFusionModel(pvd_model, pc2_model, **cfg.model.as_kwargs())
    for p in model.parameters():
        p.requires_grad_(False)
    return model
