import sys, time, torch, json
sys.path.insert(0, '.')
import bench
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.utils.procedural import fill_module_
dev = torch.device("cuda", 0)
cfg = ProjectConfig(); cfg.dataset.max_points = 1024
model = fill_module_(get_model(cfg).eval(), seed=11).to(dev)
b = next(iter(SyntheticShapes(range(1), 1, seed=5, image_size=224, num_points=1024))).to(dev)
def gpu_run():
    model._cond_cache = None
    return model.forward_sample(num_points=1024, camera=b.camera, image_rgb=b.image_rgb, mask=None, scheduler="ddpm", num_inference_steps=100)
def timed():
    torch.cuda.synchronize(); t0 = time.perf_counter(); gpu_run(); torch.cuda.synchronize(); return time.perf_counter() - t0
print("cold/warm runs:", [round(timed(), 4) for _ in range(5)])
time.sleep(12)
print("after 12 s idle:", [round(timed(), 4) for _ in range(4)])
g = model._tape_cache
print("tape entries", len(g["tape"]), "python entries", g["tape"].python_entries)
# --- does CPU-side torch work in the same process (the oracle of bench.cpu_baseline) slow the GPU loop afterwards?
a = torch.randn(2048, 2048)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    a = (a @ a).tanh()
print("after 3 s of CPU matmuls (threads=%d):" % torch.get_num_threads(), [round(timed(), 4) for _ in range(4)])
torch.set_num_threads(1)
print("after set_num_threads(1):", [round(timed(), 4) for _ in range(3)])
# --- does a C2-sized run in the same process (bench.py before c1_full) slow the C1 loop?
if "--c2" in sys.argv:
    from bdm_amd.pvd import prepare_pvd_model
    from bdm_amd.sampling import batch_streams, bdm_blending
    from bdm_amd import ops
    c2 = ProjectConfig(); c2.dataset.max_points = 4096; c2.aux_run.roll_step = 16; c2.aux_run.milestones = [1000, 968, 936]; c2.run.rng = "per_shape"
    big = fill_module_(get_model(c2).eval(), seed=1).to(dev)
    pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, dev)
    bb = next(iter(SyntheticShapes(list(range(16)), 16, seed=1, image_size=224, num_points=4096))).to(dev)
    bdm_blending(None, bb, c2, big, pvd, streams=batch_streams(c2, bb, dev, sample_idx=1))
    torch.cuda.synchronize()
    print("after a C2-sized blending run:", [round(timed(), 4) for _ in range(4)])
    from bdm_amd.profiling import KernelClassProfiler
    prof = KernelClassProfiler(every=4).install()
    bdm_blending(None, bb, c2, big, pvd, streams=batch_streams(c2, bb, dev, sample_idx=1))
    prof.remove()
    torch.cuda.synchronize()
    print("after a profiled C2 run:", [round(timed(), 4) for _ in range(4)])
    del big, pvd, bb
    import gc; gc.collect(); torch.cuda.empty_cache()
    print("after freeing the C2 models:", [round(timed(), 4) for _ in range(3)])
