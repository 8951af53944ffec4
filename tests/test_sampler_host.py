"""CPU tests of the host-side sampler logic (no GPU, no HIP calls)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ddpm_scheduler_closed_form():
    """step coefficients == the DDPM posterior q(x_{t-1} | x_t, x_0) in float64 (Ho et al. eq. 6-7)."""
    from bdm_amd.schedulers import DDPMScheduler
    s = DDPMScheduler(beta_start=1e-5, beta_end=8e-3, beta_schedule="linear", clip_sample=False)
    s.set_timesteps(1000)
    assert s.timesteps.tolist() == list(range(999, -1, -1))
    betas = np.linspace(1e-5, 8e-3, 1000, dtype=np.float32).astype(np.float64)
    abar = np.cumprod(1 - betas)
    for t in (999, 500, 1, 0):
        c = s.step_coefficients(t)
        ab_prev = abar[t - 1] if t > 0 else 1.0
        beta_t = 1 - abar[t] / ab_prev
        # the scheduler works in float32 like diffusers: 1 - abar_t cancels badly at small t
        tol = 2e-2 if t < 10 else 2e-4
        assert math.isclose(c["coef_x0"], math.sqrt(ab_prev) * beta_t / (1 - abar[t]), rel_tol=tol, abs_tol=1e-9)
        assert math.isclose(c["coef_x"], math.sqrt(1 - beta_t) * (1 - ab_prev) / (1 - abar[t]), rel_tol=tol, abs_tol=1e-9)
        assert math.isclose(c["sigma"] ** 2, max((1 - ab_prev) / (1 - abar[t]) * beta_t, 1e-20), rel_tol=2 * tol, abs_tol=1e-18)
        assert math.isclose(c["sqrt_alpha_prod"], math.sqrt(abar[t]), rel_tol=1e-5)
    s.set_timesteps(100)  # config C1: 100 steps -> leading spacing, stride 10
    assert s.timesteps.tolist()[:3] == [990, 980, 970] and s.previous_timestep(990) == 980


def test_ddpm_host_coefficients_match_oracle_step():
    from bdm_amd.schedulers import DDPMScheduler
    from oracle.ref_sampler import RefDDPM
    s, o = DDPMScheduler(beta_start=1e-5, beta_end=8e-3, clip_sample=False), RefDDPM()
    s.set_timesteps(1000)
    g = torch.Generator().manual_seed(0)
    x, eps, z = (torch.randn(2, 50, 3, generator=g) for _ in range(3))
    for t in (999, 321, 1, 0):
        c = s.step_coefficients(t)
        x0 = (x - c["sqrt_beta_prod"] * eps) / c["sqrt_alpha_prod"]
        mine = c["coef_x0"] * x0 + c["coef_x"] * x + (c["sigma"] * z if t > 0 else 0)
        assert torch.allclose(mine, o.step(eps, t, x, z), rtol=0, atol=1e-6)


def test_schedule_segment_arithmetic():
    from bdm_amd.sampling import count_forwards
    assert count_forwards() == (1000, 80, 0)
    assert count_forwards(merging=True) == (995, 75, 5)


def test_config_overrides_reference_recipe():
    from bdm_amd.config import parse_overrides
    cfg = parse_overrides(["logging.wandb_project=bdm", "run.job=sample_bdm_blending", "run.save_dir=./outputs",
                           "run.num_inference_steps=1000", "run.diffusion_scheduler=ddpm", "run.name=x", "dataset=shapenet_r2n2",
                           "dataset.root=/data", "dataset.r2n2_dir=/r2n2", "dataset.image_size=224", "dataset.category=chair",
                           "dataset.max_points=4096", "dataset.subset_ratio=0.1", "dataloader.batch_size=16",
                           "dataloader.num_workers=8", "checkpoint.resume=ckpt.pth", "aux_run.roll_step=16",
                           "aux_run.milestones=[1000,968,936,872,128,64,32,0]", "aux_run.prior_ckpt=p.pth", "aux_run.recon_ckpt=r.pth"])
    assert cfg.dataset.type == "shapenet_r2n2" and cfg.dataset.max_points == 4096 and cfg.dataloader.batch_size == 16
    assert cfg.aux_run.milestones == [1000, 968, 936, 872, 128, 64, 32, 0] and cfg.model.beta_end == 8e-3
    with pytest.raises(KeyError):
        parse_overrides(["run.no_such_key=1"])


def test_model_state_dict_layout():
    """checkpoint compatibility of the top-level model."""
    from bdm_amd.config import ProjectConfig
    from bdm_amd.model import get_model
    m = get_model(ProjectConfig())
    keys = list(m.state_dict().keys())
    assert "feature_model.model.cls_token" in keys and "feature_model.model.blocks.11.mlp.fc2.bias" in keys
    assert "point_cloud_model.model.sa_layers.1.0.voxel_layers.6.q.weight" in keys
    assert "point_cloud_model.model.classifier.2.weight" in keys
    assert m.in_channels == 390 and sum(1 for k in keys if k.startswith("point_cloud_model.model.")) == 298


def test_rasterizer_oracle_geometry():
    from bdm_amd.cameras import r2n2_camera
    from oracle.ref_sampler import owner_pixels, project_points, rasterize_bruteforce
    cam = r2n2_camera(30.0, 27.0, 1.5).packed()[0]
    pts = torch.tensor([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [5.0, 5.0, 5.0]])
    u, v, d = project_points(pts, cam)
    assert abs(float(u[0])) < 1e-6 and abs(float(v[0])) < 1e-6 and abs(float(d[0]) - 1.5) < 1e-5  # camera looks at the origin
    H = W = 32
    idx = rasterize_bruteforce(pts, cam, H, W, 0.05)
    own = owner_pixels(pts, cam, H, W, 0.05)
    assert int((idx == 0).sum()) >= 1 and int((idx == 1).sum()) == 0  # equal depth: the EARLIER point wins
    assert own[1].item() == -1 and own[0].item() == int(torch.nonzero(idx.reshape(-1) == 0).max())


def test_shard_indices_cover_and_balance():
    from bdm_amd.distributed import shard_indices
    for n, w in [(128, 8), (16, 1), (10, 4), (3, 8)]:
        parts = [shard_indices(n, r, w) for r in range(w)]
        assert sum(parts, []) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_two_rank_gloo_sharding_equivalence(tmp_path):
    """world_size 2 over gloo: per-shape noise streams and the result gather are rank-count invariant."""
    script = tmp_path / "worker.py"
    script.write_text(f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from bdm_amd.data import SyntheticShapes, shape_generator
from bdm_amd.distributed import init_from_env, shard_indices, gather_clouds, barrier, max_over_ranks, per_rank_values
rank, local_rank, world = init_from_env(backend="gloo")
N, TOTAL = 64, 5
idx = shard_indices(TOTAL, rank, world)
# stand-in for a trajectory: a deterministic function of the per-shape streams only
local = torch.stack([torch.randn(N, 3, generator=shape_generator(42, j)) for j in idx]) if idx else torch.zeros(0, N, 3)
batches = list(SyntheticShapes(idx, 2, seed=42, image_size=8, num_points=N))
assert sum(b.image_rgb.shape[0] for b in batches) == len(idx)
barrier()
full = gather_clouds(local, TOTAL, rank, world)
ref = torch.stack([torch.randn(N, 3, generator=shape_generator(42, j)) for j in range(TOTAL)])
assert torch.equal(full, ref), "gathered result differs from the single-rank result"
assert max_over_ranks(float(rank), torch.device("cpu")) == world - 1
assert per_rank_values(10.0 + rank, torch.device("cpu")) == [10.0 + r for r in range(world)]   # bench.py's per_rank_s
print("OK", rank)
""")
    import socket
    with socket.socket() as sk:  # a free port (a fixed one collides with the TIME_WAIT of a run just before)
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("OK") == 2


def test_worker_pins_itself_to_its_gpus_numa_node(tmp_path):
    """bench.py's workers pin themselves to the host cores of the NUMA node their GPU hangs off (distributed.pin_to_gpu_numa_node),
    read from sysfs: exercised on a fake tree (two amdgpu render nodes on nodes 0 / 1, one foreign device), in a child process so
    that this test process keeps its affinity.  An unreadable topology is reported, never raised."""
    sysfs = tmp_path / "sys"
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    nodes = {0: allowed[:half], 1: allowed[half:] or allowed[:half]}
    for n, cpus in nodes.items():
        d = sysfs / "devices" / "system" / "node" / f"node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    for i, (pci, vendor, node) in enumerate([("0000:05:00.0", "0x1002", 0), ("0000:85:00.0", "0x1002", 1), ("0000:01:00.0", "0x10de", 0)]):
        dev = sysfs / "devices" / "pci" / pci
        dev.mkdir(parents=True)
        (dev / "vendor").write_text(vendor + "\n")
        (dev / "numa_node").write_text(f"{node}\n")
        rd = sysfs / "class" / "drm" / f"renderD{128 + i}"
        rd.mkdir(parents=True)
        os.symlink(dev, rd / "device")
    code = f"""
import os, sys, json
sys.path.insert(0, {ROOT!r})
from bdm_amd.distributed import pin_to_gpu_numa_node
out = []
for lr in (0, 1, 2):
    info = pin_to_gpu_numa_node(lr, sysfs={str(sysfs)!r})
    out.append((info, sorted(os.sched_getaffinity(0))))
    os.sched_setaffinity(0, {allowed!r})
out.append((pin_to_gpu_numa_node(0, sysfs="/nonexistent"), None))
# KFD topology present: HIP order = KFD GPU-node order (here the REVERSE of the PCI order), CPU nodes (simd_count 0) skipped
for nid, (simd, minor) in enumerate([(0, -1), (256, 129), (256, 128)]):
    d = os.path.join({str(sysfs)!r}, "class/kfd/kfd/topology/nodes", str(nid))
    os.makedirs(d)
    open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\\nsimd_count {{simd}}\\ndrm_render_minor {{minor}}\\n")
out.append((pin_to_gpu_numa_node(0, sysfs={str(sysfs)!r}), None))
os.sched_setaffinity(0, {allowed!r})
os.environ["HIP_VISIBLE_DEVICES"] = "1"      # ordinal into the KFD order -> the 0000:05 device
out.append((pin_to_gpu_numa_node(0, sysfs={str(sysfs)!r}), None))
os.sched_setaffinity(0, {allowed!r})
os.environ["ROCR_VISIBLE_DEVICES"] = "GPU-deadbeef"   # cannot be interpreted: no pinning, no guess
out.append((pin_to_gpu_numa_node(0, sysfs={str(sysfs)!r}), sorted(os.sched_getaffinity(0))))
print(json.dumps(out))
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    import json
    r0, r1, r2, r3, k0, k1, k2 = json.loads(res.stdout.strip().splitlines()[-1])
    assert k0[0]["pci"] == "0000:85:00.0" and k0[0]["order"] == "kfd" and k0[0]["numa_node"] == 1
    assert k1[0]["pci"] == "0000:05:00.0" and k1[0]["numa_node"] == 0
    assert k2[0]["numa_node"] is None and "VISIBLE_DEVICES" in k2[0]["why"] and k2[1] == allowed
    assert r0[0]["order"] == "pci"
    assert r0[0]["numa_node"] == 0 and r0[1] == nodes[0] and r0[0]["pci"] == "0000:05:00.0"
    assert r1[0]["numa_node"] == 1 and r1[1] == nodes[1]
    assert r2[0]["numa_node"] is None and r2[1] == allowed          # only two amdgpu devices: rank 2 is left alone
    assert r3[0]["numa_node"] is None


def test_windowed_rasterizer_equals_bruteforce():
    from bdm_amd.cameras import r2n2_camera
    from oracle import ref_sampler as R
    g = torch.Generator().manual_seed(4)
    for H, radius, n in [(32, 0.05, 300), (64, 0.02, 800)]:
        cam = r2n2_camera(75.0, 28.0, 1.5).packed()[0]
        pts = torch.randn(n, 3, generator=g) * 0.3
        pts[:5] = pts[5:10]
        a = R.rasterize_bruteforce(pts, cam, H, H, radius)
        b = R.rasterize_windowed(pts, cam, H, H, radius)
        assert torch.equal(a, b)
        feat = torch.randn(6, H, H, generator=g)
        R.FAST_RASTER = False
        slow = R.surface_projection(pts, cam, feat, radius)
        R.FAST_RASTER = True
        assert torch.equal(slow, R.surface_projection(pts, cam, feat, radius))


def test_ddim_scheduler_closed_form_and_schedule_mapping():
    from bdm_amd.config import ProjectConfig
    from bdm_amd.sampling import _schedule
    from bdm_amd.schedulers import DDIMScheduler
    s = DDIMScheduler(beta_start=1e-5, beta_end=8e-3, clip_sample=False)
    s.set_timesteps(64)  # the reference's DDIM recipe: 64 recon steps, stride 15
    assert s.timesteps.tolist()[:3] == [945, 930, 915] and s.timesteps.tolist()[-1] == 0
    abar = np.cumprod(1 - np.linspace(1e-5, 8e-3, 1000, dtype=np.float32).astype(np.float64))
    c = s.step_coefficients(945)
    assert math.isclose(c["coef_x0"], math.sqrt(abar[930]), rel_tol=1e-5)
    assert math.isclose(c["coef_eps"], math.sqrt(1 - abar[930]), rel_tol=1e-4) and c["sigma"] == 0.0
    c0 = s.step_coefficients(0)
    assert math.isclose(c0["coef_x0"], 1.0) and abs(c0["coef_eps"]) < 1e-6  # last step lands on x0
    cfg = ProjectConfig()
    cfg.run.diffusion_scheduler, cfg.aux_run.roll_step, cfg.aux_run.milestones = "ddim", 1, [64, 62, 60, 56, 8, 4, 2, 0]
    roll, ms, proll, pms, times = _schedule(cfg)
    assert (proll, pms, times) == (16, [1000, 968, 937, 875, 125, 62, 31, 0], 7)  # main_blending.py:214-218


def test_screen_space_and_pix3d_cameras():
    """in_ndc=False intrinsics convert to NDC with pytorch3d's rule; the Pix3D adapter composes crop + resize with K."""
    from bdm_amd.cameras import PerspectiveCameras, pix3d_camera
    c = PerspectiveCameras(focal_length=[[300.0, 300.0]], principal_point=[[100.0, 140.0]], in_ndc=False, image_size=(224, 224))
    assert torch.allclose(c.focal_length, torch.tensor([[300 * 2 / 224.0] * 2]))
    assert torch.allclose(c.principal_point, torch.tensor([[-(100 - 112) * 2 / 224.0, -(140 - 112) * 2 / 224.0]]))
    # a 640x480 photo, object box (200,100)-(440,420): crop half-side 160 around (320, 260), resized to 224
    cam = pix3d_camera(torch.eye(3).numpy(), [0.0, 0.0, 2.0], [0.0, 0.0, 0.0], 1.0, (640, 480), (200, 100, 440, 420), 35.0)
    f, s = 35.0 * 640 / 32, 224 / 320.0
    assert torch.allclose(cam.focal_length, torch.tensor([[s * f * 2 / 224] * 2]))
    tx, ty = s * (320 - 160), s * (240 - 100)
    assert torch.allclose(cam.principal_point, torch.tensor([[-(tx - 112) * 2 / 224, -(ty - 112) * 2 / 224]]), atol=1e-6)
    # unit-std normalisation scales the rotation, the mean shifts the translation; OpenCV -> pytorch3d axis swap
    cam2 = pix3d_camera(torch.eye(3).numpy(), [0.0, 0.0, 2.0], [0.1, 0.0, 0.0], 2.0, (640, 480), (200, 100, 440, 420), 35.0)
    assert torch.allclose(cam2.T, torch.tensor([[0.1, 0.0, 2.0]]))
    assert torch.allclose(cam2.R[0], 2.0 * torch.tensor([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]]).T)


def test_pndm_timestep_layout_and_exactness_on_a_consistent_model():
    """PNDM (the reference's schedulers_map['pndm'], model.py:61): 50 inference steps -> 12 Runge-Kutta stages + 47 linear
    multistep steps = 59 network evaluations; for a model that always predicts the noise consistent with ONE fixed x0 the
    transfer formula is exact, so the chain must land on x0 (up to final_alpha_cumprod = alphas_cumprod[0] != 1)."""
    from oracle.ref_sampler import RefPNDM
    s = RefPNDM()
    s.set_timesteps(50)
    assert len(s.timesteps) == 59
    assert s.timesteps[:12].tolist() == [980, 970, 970, 960, 960, 950, 950, 940, 940, 930, 930, 920]
    assert s.timesteps[12:15].tolist() == [920, 900, 880] and s.timesteps[-1] == 0
    g = torch.Generator().manual_seed(0)
    x0, z = torch.randn(2, 64, 3, generator=g), torch.randn(2, 64, 3, generator=g)
    a = s.ac[980]
    x = a.sqrt() * x0 + (1 - a).sqrt() * z

    def eps_of(x, t):   # the noise that explains x at level t given the fixed x0
        return (x - s.ac[t].sqrt() * x0) / (1 - s.ac[t]).sqrt()
    for t in s.timesteps:
        # PRK stages evaluate the network at the stage's own (t, sample) pair, exactly as the reference's loop does
        x = s.step(eps_of(x, int(t)), int(t), x)
    assert float((x - s.final.sqrt() * x0).norm() / x0.norm()) < 0.05


def test_pndm_product_scheduler_layout():
    from bdm_amd.schedulers import make_schedulers_map
    m = make_schedulers_map(beta_start=1e-5, beta_end=8e-3, beta_schedule="linear")
    p = m["pndm"]
    p.set_timesteps(50)
    assert len(p.timesteps) == 59 and p.timesteps[:4].tolist() == [980, 970, 970, 960]


def test_distance_transform_chamfer_vs_exact():
    """compute_distance_transform (model_utils.py:13-21 over cv2's 3x3 DIST_L2 chamfer metric): zero on the mask, grows away
    from it, within the chamfer-3x3 metric's known relative error (< 6 %) of the exact Euclidean transform, scaled by
    image_size / 2 and clipped to [0, 1]."""
    from scipy import ndimage
    from bdm_amd.model import compute_distance_transform
    g = torch.Generator().manual_seed(0)
    m = torch.zeros(2, 1, 48, 48, dtype=torch.bool)
    m[0, 0, 10:20, 12:30] = True
    m[1, 0][torch.rand(48, 48, generator=g) > 0.97] = True
    dt = compute_distance_transform(m)
    assert dt.shape == (2, 1, 48, 48) and float(dt.min()) == 0.0 and float(dt.max()) <= 1.0
    assert bool((dt[m] == 0).all())
    for b in range(2):
        exact = ndimage.distance_transform_edt(~m[b, 0].numpy()) / 24.0
        got = dt[b, 0].numpy()
        sel = exact < 0.95                                     # below the clip
        assert np.all(np.abs(got[sel] - exact[sel]) <= 0.06 * exact[sel] + 1e-6)
    # horizontal / vertical neighbours are exactly a = 0.955 pixels away in this metric
    one = torch.zeros(1, 1, 9, 9, dtype=torch.bool); one[0, 0, 4, 4] = True
    d1 = compute_distance_transform(one)[0, 0] * (9 / 2)
    assert abs(float(d1[4, 5]) - 0.955) < 1e-4 and abs(float(d1[5, 5]) - 1.3693) < 1e-4 and abs(float(d1[4, 6]) - 1.91) < 1e-4


def test_mask_conditioning_changes_the_input_width():
    from bdm_amd.config import ProjectConfig
    from bdm_amd.model import get_model
    cfg = ProjectConfig()
    base = get_model(cfg).in_channels
    cfg.model.use_mask, cfg.model.use_distance_transform = True, True
    m = get_model(cfg)
    assert (base, m.in_channels) == (390, 392) and m.point_cloud_model.model.in_channels == 392
    cfg.model.use_distance_transform = False
    assert get_model(cfg).in_channels == 391
