"""Eight REPLAYED PC^2 reverse steps (launch tape, B=16, N=4096) between two marker kernels, for `rocprofv3 --kernel-trace`:
tools/trace_summary.py / trace_timeline.py then show the launches of those steps without any host-side pacing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.utils.procedural import fill_module_
B, N = int(os.environ.get("TB", 16)), int(os.environ.get("TN", 4096))
from bdm_amd import pvcnn as _pv
for _k in ("SIDE_PLAN", "SIDE_NN", "DEFER_CHAIN"):          # experiments: TRACE_SIDE_PLAN=0 etc.
    if os.environ.get("TRACE_" + _k) is not None:
        setattr(_pv, _k, os.environ["TRACE_" + _k] == "1")
if os.environ.get("TRACE_NO_WAIT") == "1":   # UNSAFE (stale inputs, timing experiment only): the main stream never waits for the side streams
    from bdm_amd import tape as _tape
    _tape.wait_event = lambda event: None
cfg = ProjectConfig(); cfg.dataset.max_points = N
model = fill_module_(get_model(cfg).eval(), seed=1).cuda()
b = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda") * 0.5
sched = model.schedulers_map["ddpm"]; sched.set_timesteps(1000)
model._denoise_loop(x, b.camera, b.image_rgb, None, sched, list(range(999, 987, -1)))
torch.cuda.synchronize()
marker = torch.zeros(7, device="cuda")
torch.cumsum(marker, 0); torch.cuda.synchronize()
model._denoise_loop(x, b.camera, b.image_rgb, None, sched, list(range(987, 979, -1)))
torch.cuda.synchronize()
torch.cumsum(marker, 0); torch.cuda.synchronize()
assert model._tape_cache["tape"] is not None
