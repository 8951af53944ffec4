"""Would two concurrent half-batches beat one full batch?  The main hardware queue of a B = 16 forward is idle ~11 % of the time
between dependent launches (profiles/r02_forward_launches.txt); two independent lanes of 8 shapes on two streams can fill those gaps
-- if the kernels do not lose more at half size.  Records the PC^2 step (conditioning + denoiser) as launch tapes on separate streams
and times interleaved native replays against the full-batch tape.   usage: lanes_probe.py [B] [N] [steps] [lanes]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bdm_amd.model as M
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.utils.procedural import fill_module_

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps, lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 40, int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = ProjectConfig(); cfg.dataset.max_points = N
model = fill_module_(M.get_model(cfg).eval(), seed=1).cuda()
M.TAPE_STEPS = "1"


def record(batch, x, stream):
    """-> the tape cache of a recorded step for this (half-)batch on `stream`"""
    model._tape_cache = None
    model._cond_cache = None
    with torch.cuda.stream(stream):
        model.interaction_sample(x.clone(), batch.camera, batch.image_rgb, None, start_time=900, end_time=890)
    torch.cuda.synchronize()
    g = model._tape_cache
    assert g["tape"] is not None and g["off"] is None, g["off"]
    return g


def host_time(tapes, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for g in tapes:
            g["tape"].replay()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return dt * 1e3


def run(tapes, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        for g in tapes:
            g["tape"].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


full_batch = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda")
s0 = torch.cuda.Stream()
g_full = record(full_batch, x, s0)
print(f"B={B} N={N}: full-batch tape: {len(g_full['tape'])} entries, {g_full['tape'].python_entries} python entries; "
      f"host enqueue {host_time([g_full]):.2f} ms / forward; {run([g_full], steps):.2f} ms / forward")
per = B // lanes
gs = []
for i in range(lanes):
    hb = next(iter(SyntheticShapes(range(i * per, (i + 1) * per), per, num_points=N))).to("cuda")
    gs.append(record(hb, x[i * per:(i + 1) * per].contiguous(), torch.cuda.Stream()))
print(f"{lanes} lanes of {per}: host enqueue {host_time(gs):.2f} ms / step; {run(gs, steps):.2f} ms / step (all {B} shapes)")
print(f"one lane of {per} alone: {run(gs[:1], steps):.2f} ms / forward")
