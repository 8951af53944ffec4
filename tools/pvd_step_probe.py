"""PVD prior step at B=16, N=4096: eager loop time per step vs the GPU time of its kernels (is the prior's loop host-bound?).
python tools/pvd_step_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.pvd import prepare_pvd_model, generate_pvd_xyz
B, N = 16, 4096
pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cuda")
x = torch.randn(B, 3, N, device="cuda") * 0.5
def run(steps):
    return generate_pvd_xyz(pvd, x, 500, 500 - steps)
run(4); torch.cuda.synchronize()
t0 = time.perf_counter(); run(32); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 32 * 1e3
# GPU time of a single forward measured with events around back-to-back forwards (host ahead? then event time ~ GPU time)
t = torch.full((B,), 400, dtype=torch.int64, device="cuda")
for _ in range(3): pvd.model(x, t)
torch.cuda.synchronize()
h0 = time.perf_counter()
for _ in range(10): pvd.model(x, t)
host = (time.perf_counter() - h0) / 10 * 1e3
torch.cuda.synchronize()
print(f"PVD loop {wall:.2f} ms per step | host enqueue of one forward {host:.2f} ms (if below the step time, the loop is GPU-bound)")
