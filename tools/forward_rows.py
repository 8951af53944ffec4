"""Per (C-ABI function, shape) device time of the PC2 forward at the bench's shape (B=16, N=4096), measured with the live
profiler of bench.py (HIP events around EVERY launch, `iters` forwards).  usage: forward_rows.py [filter] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdm_amd.config import ProjectConfig
from bdm_amd.model import get_model
from bdm_amd.data import SyntheticShapes
from bdm_amd.utils.procedural import fill_module_
from bdm_amd.profiling import KernelClassProfiler

flt = sys.argv[1] if len(sys.argv) > 1 else ""
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, N = 16, 4096
cfg = ProjectConfig()
cfg.dataset.max_points = N
model = fill_module_(get_model(cfg).eval(), seed=1).cuda()
batch = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda") * 0.5
t = torch.full((B,), 500, dtype=torch.int64, device="cuda")
with torch.no_grad():
    xin = model.get_input_with_conditioning(x, batch.camera, batch.image_rgb, None, t)
    for _ in range(3):
        model.point_cloud_model(xin, t)
    torch.cuda.synchronize()
    prof = KernelClassProfiler(every=1).install()
    for _ in range(iters):
        model.point_cloud_model(xin, t)
    torch.cuda.synchronize()
    prof.remove()
rows, classes = prof.table()
tot = 0.0
for r in rows:
    if flt in r["function"]:
        per_fwd = r["est_total_ms"] * 1e3 / iters
        tot += per_fwd
        extra = ""
        if "pointwise" in r["function"] and len(r["shape"]) >= 4:  # (b, m, k, n): GEMM rate and the bytes of one read of x + one write of y
            b_, m_, k_, n_ = r["shape"][:4]
            extra = f"  {2.0 * b_ * m_ * k_ * n_ / r['avg_us'] / 1e6:6.1f} TF/s  {4.0 * b_ * n_ * (k_ + m_) / r['avg_us'] / 1e3:7.1f} GB/s"
        print(f"{r['function']:34s} {str(tuple(r['shape'])):28s} x{r['calls'] // iters:3d}  {r['avg_us']:8.1f} us  {per_fwd:8.1f} us/forward{extra}")
print(f"total {tot:.1f} us/forward over {sum(r['calls'] for r in rows if flt in r['function']) // iters} launches")
if not flt:
    for c in classes:
        print(f"  {c['class']:40s} {100 * c['share']:5.1f} %  {c['kernel_ms'] * 1e3 / iters:8.1f} us  x{c['launches'] // iters}")
