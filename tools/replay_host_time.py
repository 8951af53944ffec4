"""Is the host ahead of the GPU in the replayed reverse loop?  Host time to ENQUEUE k replayed PC^2 steps (the loop returns) against the
time until the GPU has finished them.  python tools/replay_host_time.py [B] [N] [steps]

GATED=1: the K steps are enqueued into a main stream that is BLOCKED behind a long spin kernel (torch.cuda._sleep,
calibrated to ~3x the expected GPU time of the K steps), i.e. into an idle GPU with room in every queue: the time until the loop
returns is then the host's own cost of a replayed step (tape walk, ctypes, event records), free of back-pressure from a full queue.
K is kept small (default 10 when gated) so that the ~2300 packets fit the hardware queues while the gate is shut."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.utils.procedural import fill_module_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
GATED = os.environ.get("GATED") == "1"
K = int(sys.argv[3]) if len(sys.argv) > 3 else (10 if GATED else 40)
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
cycles_per_ms = None
if GATED:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000); torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
    cycles_per_ms = 20_000_000 / e0.elapsed_time(e1)
    print(f"spin kernel: {cycles_per_ms:.0f} cycles per ms", flush=True)
for rep in range(3):
    if GATED:
        gate_ms = min(3 * 8.0 * K, 1500.0)
        torch.cuda._sleep(int(gate_ms * cycles_per_ms))   # the main stream (and every stream that forks from it) waits behind this
    t0 = time.perf_counter()
    model._denoise_loop(x, b.camera, b.image_rgb, None, sched, list(range(987, 987 - K, -1)))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if GATED:
        print(f"B={B} N={N}: {K} replayed steps into a gated (idle) GPU, gate {gate_ms:.0f} ms: host enqueue {(t1 - t0) / K * 1e3:.3f} ms / step "
              f"(host returned {'BEFORE' if (t1 - t0) * 1e3 < gate_ms else 'AFTER'} the gate opened)", flush=True)
        continue
    print(f"B={B} N={N}: {K} replayed steps: host returned after {(t1 - t0) / K * 1e3:.3f} ms / step, GPU done after {(t2 - t0) / K * 1e3:.3f} ms / step", flush=True)
