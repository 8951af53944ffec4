"""Would the 32^3 PVConvs run faster as two sequential half batches (intermediates of 67 MB instead of 134 MB: inside the 256 MB
memory-side cache)?  Times PVConv modules at B = 16 vs twice B = 8 on the same inputs.  python tools/pvconv_split_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.modules import PVConv
from bdm_amd.utils.procedural import fill_module_
from bdm_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout, r, n in [(32, 32, 32, 4096), (64, 64, 32, 4096), (192, 64, 32, 4096), (128, 128, 16, 1024), (64, 64, 16, 1024)]:
    pv = fill_module_(PVConv(cin, cout, 3, resolution=r, with_se=True, with_se_relu=True).eval(), seed=cin + r).cuda()
    g = torch.Generator().manual_seed(n)
    f, c = torch.randn(16, cin, n, generator=g).cuda(), (torch.randn(16, 3, n, generator=g) * 0.5).cuda()
    te = torch.zeros(16, 8, n, device="cuda")
    halves = [(f[:8].contiguous(), c[:8].contiguous(), te[:8].contiguous()), (f[8:].contiguous(), c[8:].contiguous(), te[8:].contiguous())]
    def full():
        ops.clear_plan_cache(); return pv((f, c, te))[0]
    def split():
        out = []
        for h in halves:
            ops.clear_plan_cache(); out.append(pv(h)[0])
        return out
    a = full(); b2 = torch.cat(split(), 0)
    print(f"PVConv {cin:3d}->{cout:3d} r={r:2d} n={n}: B=16 {t(full):7.1f} us | 2 x B=8 {t(split):7.1f} us | equal {torch.equal(a, b2)}", flush=True)
