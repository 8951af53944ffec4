"""Times one denoiser forward (and the per-step conditioning) on the HIP path. Usage: python tools/time_forward.py [B] [N] [iters]"""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.pvd import prepare_pvd_model
from bdm_amd.utils.procedural import fill_module_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = ProjectConfig()
cfg.dataset.max_points = N
model = fill_module_(get_model(cfg).eval(), seed=1).cuda()
pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cuda")
batch = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda") * 0.5
t = torch.full((B,), 500, dtype=torch.int64, device="cuda")


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


cond = lambda: model.get_input_with_conditioning(x, batch.camera, batch.image_rgb, None, t)
xin = cond()
print(f"B={B} N={N}")
print(f"conditioning      {timeit(cond, iters):9.2f} ms")
print(f"PC2 forward       {timeit(lambda: model.point_cloud_model(xin, t), iters):9.2f} ms")
xp = x.transpose(1, 2).contiguous()
print(f"PVD forward       {timeit(lambda: pvd.model(xp, t), iters):9.2f} ms")
step = lambda: model.interaction_sample(x, batch.camera, batch.image_rgb, None, start_time=500, end_time=499)
print(f"full PC2 step     {timeit(step, iters):9.2f} ms")

# CPU-side enqueue cost of one PC2 forward (no synchronisation inside the loop)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    model.point_cloud_model(xin, t)
cpu_ms = (time.perf_counter() - t0) / iters * 1e3
torch.cuda.synchronize()
print(f"CPU enqueue / forward {cpu_ms:7.2f} ms")
