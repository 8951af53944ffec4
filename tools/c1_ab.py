"""C1 (one shape, N = 1024, 100 steps) wall time, six runs.  C1_SE_IN_DEVOX=1: SE block's FC layers inside the devoxelisation kernel
(one launch less per PVConv).  python tools/c1_ab.py"""
import os, sys, time, torch
sys.path.insert(0, '.')
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.utils.procedural import fill_module_
from bdm_amd import modules as _m
if os.environ.get("C1_SE_IN_DEVOX") == "1":
    _m.PVConv.se_in_devox = True
dev = torch.device("cuda", 0)
cfg = ProjectConfig(); cfg.dataset.max_points = 1024
model = fill_module_(get_model(cfg).eval(), seed=11).to(dev)
b = next(iter(SyntheticShapes(range(1), 1, seed=5, image_size=224, num_points=1024))).to(dev)
def gpu_run():
    model._cond_cache = None
    return model.forward_sample(num_points=1024, camera=b.camera, image_rgb=b.image_rgb, mask=None, scheduler="ddpm", num_inference_steps=100)
def timed():
    torch.cuda.synchronize(); t0 = time.perf_counter(); gpu_run(); torch.cuda.synchronize(); return time.perf_counter() - t0
print("C1 runs:", [round(timed(), 4) for _ in range(6)], "tape entries", len(model._tape_cache["tape"]))
