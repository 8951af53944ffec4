"""Repeats one mini BDM-Blending trajectory (per-shape streams, fixed seed) and reports whether all repetitions are
bit-identical -- optionally while a second process keeps the GPU busy (the sporadic 1-rank vs 2-rank difference of
tests/test_hip_cli.py only showed up with two processes on one GPU).  Usage: determinism_probe.py [reps] [load]"""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

if len(sys.argv) > 1 and sys.argv[1] == "--load":
    x = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(50):
            x = torch.tanh(x @ x * 1e-3)
        torch.cuda.synchronize()
    sys.exit(0)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
load = len(sys.argv) > 2 and sys.argv[2] == "load"
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.pvd import prepare_pvd_model
from bdm_amd.sampling import batch_streams, bdm_blending
from bdm_amd.utils.procedural import fill_module_
cfg = ProjectConfig()
cfg.dataset.max_points, cfg.run.rng = 1024, "per_shape"
cfg.aux_run.milestones, cfg.aux_run.roll_step = [1000, 998, 996, 995], 1
model = fill_module_(get_model(cfg).eval(), seed=42).cuda()
pvd = prepare_pvd_model({"model": "procedural:1", "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cuda")
batch = next(iter(SyntheticShapes([2, 3], 2, seed=42, num_points=1024))).to("cuda")
proc = subprocess.Popen([sys.executable, __file__, "--load", str(8 + 3 * reps)]) if load else None
time.sleep(3 if load else 0)
outs = []
mode = os.environ.get("PROBE", "traj")
xt = torch.randn(2, 1024, 3, generator=torch.Generator().manual_seed(3)).cuda() * 0.5
tt = torch.full((2,), 500, dtype=torch.int64, device="cuda")
for i in range(reps):
    model._cond_cache = None
    if mode == "traj":
        outs.append(bdm_blending(None, batch, cfg, model, pvd, streams=batch_streams(cfg, batch, "cuda", 0)).points_padded().cpu())
    elif mode == "cond":
        outs.append(model.get_input_with_conditioning(xt, batch.camera, batch.image_rgb, None, tt).cpu())
    elif mode == "vit":
        outs.append(model.conditioning_image(batch.image_rgb)[0].cpu())
    elif mode == "pc2":
        if i == 0:
            xin = model.get_input_with_conditioning(xt, batch.camera, batch.image_rgb, None, tt)
        outs.append(model.point_cloud_model(xin, tt).cpu())
    elif mode == "pvd":
        outs.append(pvd.model(xt.transpose(1, 2).contiguous(), tt).cpu())
if proc:
    proc.wait()
bad = [i for i in range(1, reps) if not torch.equal(outs[i], outs[0])]
print(f"mode={mode} env={ {k: v for k, v in os.environ.items() if k.startswith('BDM_')} } load={load}: {reps} repetitions, "
      f"{len(bad)} differ from the first" + (f" (max |diff| {max(float((outs[i] - outs[0]).abs().max()) for i in bad):.2e})" if bad else ""))
