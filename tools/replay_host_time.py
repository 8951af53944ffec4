"""Is the host ahead of the GPU in the replayed reverse loop?  Host time to ENQUEUE k replayed PC^2 steps (the loop returns) against the
time until the GPU has finished them.  python tools/replay_host_time.py [B] [N] [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.utils.procedural import fill_module_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
from bdm_amd import pvcnn as _pv
for _k in ("SIDE_PLAN", "SIDE_NN", "DEFER_CHAIN"):          # experiments: TRACE_DEFER_CHAIN=0 etc. (as tools/trace_step.py)
    if os.environ.get("TRACE_" + _k) is not None:
        setattr(_pv, _k, os.environ["TRACE_" + _k] == "1")
if os.environ.get("TRACE_NO_WAIT") == "1":   # UNSAFE (stale inputs, timing experiment only)
    from bdm_amd import tape as _tape
    _tape.wait_event = lambda event: None
cfg = ProjectConfig(); cfg.dataset.max_points = N
model = fill_module_(get_model(cfg).eval(), seed=1).cuda()
b = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda") * 0.5
sched = model.schedulers_map["ddpm"]; sched.set_timesteps(1000)
model._denoise_loop(x, b.camera, b.image_rgb, None, sched, list(range(999, 987, -1)))
torch.cuda.synchronize()
assert model._tape_cache["tape"] is not None
for rep in range(3):
    t0 = time.perf_counter()
    model._denoise_loop(x, b.camera, b.image_rgb, None, sched, list(range(987, 987 - K, -1)))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} N={N}: {K} replayed steps: host returned after {(t1 - t0) / K * 1e3:.3f} ms / step, GPU done after {(t2 - t0) / K * 1e3:.3f} ms / step", flush=True)
