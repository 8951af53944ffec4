"""Times the PC^2 reverse loop (interaction_sample) eager vs hipGraph replay. Usage: python tools/time_loop.py B N steps"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bdm_amd.model as M
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.utils.procedural import fill_module_

B, N, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = ProjectConfig(); cfg.dataset.max_points = N
model = fill_module_(M.get_model(cfg).eval(), seed=1).cuda()
batch = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda")
M.TAPE_STEPS = "0"
for graph in (False, True, False, "tape", "tape"):
    M.GRAPH_STEPS = graph is True
    M.TAPE_STEPS = "1" if graph == "tape" else "0"
    model.interaction_sample(x.clone(), batch.camera, batch.image_rgb, None, start_time=900, end_time=890)  # warm / capture
    torch.cuda.synchronize(); t0 = time.perf_counter(); c0 = time.process_time()
    model.interaction_sample(x.clone(), batch.camera, batch.image_rgb, None, start_time=800, end_time=800 - steps)
    cpu_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B} N={N} graph={graph}: {dt / steps * 1e3:.2f} ms / step (host loop returned after {cpu_enq / steps * 1e3:.2f} ms / step, "
          f"process CPU {(time.process_time() - c0) / steps * 1e3:.2f} ms / step)")
