"""GPU time of one replayed PC^2 reverse step per SHAPE as a function of the batch size (N = 4096): would the 16 shapes of C2 run
faster as sequential sub-batches (smaller intermediates: more of them stay in the 256 MB memory-side cache)?  python tools/step_vs_batch.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.utils.procedural import fill_module_
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = ProjectConfig(); cfg.dataset.max_points = N
model = fill_module_(get_model(cfg).eval(), seed=1).cuda()
for B in (2, 4, 8, 16, 32):
    b = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
    x = torch.randn(B, N, 3, device="cuda") * 0.5
    model._cond_cache = None
    sched = model.schedulers_map["ddpm"]; sched.set_timesteps(1000)
    ts = list(range(999, 899, -1))
    model._denoise_loop(x, b.camera, b.image_rgb, None, sched, ts[:12])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model._denoise_loop(x, b.camera, b.image_rgb, None, sched, ts)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / len(ts)
    print(f"B={B:2d} N={N}: {ms:7.3f} ms per step = {ms / B:6.3f} ms per shape-step", flush=True)
