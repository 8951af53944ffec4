"""First-convolution microbenchmark at the bench's layer shapes (B=16): batched GEMM + gather (bf16x6 / fp16x3, sparse_conv.hip,
sparse_conv_h2.hip) vs the output-stationary implicit GEMM with tap skipping (sparse_conv_os.hip).  Clouds: the FPS chain of a
random cloud, as the denoiser sees them.   usage: sparse_bench.py [B] [cloud sigma]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops, _lib as L
from bdm_amd import functional as F


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(0)
clouds = {4096: (torch.randn(B, 3, 4096, generator=g) * 0.5).cuda()}
if len(sys.argv) > 2 and sys.argv[2] == "heavy":   # heavier-tailed cloud (sparser grid)
    clouds[4096] = clouds[4096] * torch.randn(B, 1, 4096, generator=g).abs().cuda()
for m in (1024, 256, 64):
    clouds[m] = F.furthest_point_sample(clouds[m * 4], m)
LAYERS = [("SA0.0", 390, 32, 32, 4096), ("SA0.1", 32, 32, 32, 4096), ("SA1.0", 128, 64, 16, 1024), ("SA2.0", 192, 128, 8, 256),
          ("FP0.x", 256, 256, 8, 64), ("FP1.x", 256, 256, 8, 256), ("FP2.x", 128, 128, 16, 1024), ("FP3.x", 64, 64, 32, 4096)]
MULT = {"SA0.0": 1, "SA0.1": 1, "SA1.0": 1, "SA2.0": 1, "FP0.x": 3, "FP1.x": 3, "FP2.x": 2, "FP3.x": 2}
tot_old = tot_new = tot_os = 0.0
lib = L.lib()
for name, cin, cout, r, n in LAYERS:
    pts = clouds[n]
    ops.clear_plan_cache()
    plan = ops.voxel_plan(pts, r)
    f = torch.randn(B, cin, n, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    bias = torch.zeros(cout).cuda()
    w_old, w_new = ops.sparse_conv_pack_s3(w), ops.sparse_conv_pack_h2(w)
    occ = float(plan.n_occ.float().mean())
    a = ops.sparse_first_conv_planned(f, plan, w_old, bias, cout)
    b = ops.sparse_first_conv_planned(f, plan, w_new, bias, cout)
    err = float((a - b).norm() / a.norm())
    t_old = t(lambda: ops.sparse_first_conv_planned(f, plan, w_old, bias, cout))
    t_new = t(lambda: ops.sparse_first_conv_planned(f, plan, w_new, bias, cout))
    w_os = ops.sparse_conv_pack_os(w)[1:]
    c = ops.sparse_first_conv_os(f, plan, w_os, bias, cout)
    err_os = float((c - b).norm() / b.norm())
    t_os = t(lambda: ops.sparse_first_conv_os(f, plan, w_os, bias, cout))
    t_os_gn = t(lambda: ops.sparse_first_conv_os(f, plan, w_os, bias, cout, gn_groups=8, compact=True))
    t_k = t_new
    tot_old += MULT[name] * t_old; tot_new += MULT[name] * t_new; tot_os += MULT[name] * t_os_gn
    fl = 2 * occ * 27 * cin * cout * B
    print(f"{name} {cin:4d}->{cout:4d} r={r:2d} n={n:5d} n_occ={occ:7.1f}  bf16x6 {t_old:7.1f} us  fp16x3 {t_new:7.1f} us ("
          f"{fl / t_k / 1e6:6.1f} TF/s alg.)  rel diff {err:.1e} | output-stationary, dense out {t_os:7.1f} us; compact out + GN stats {t_os_gn:7.1f} us; rel diff {err_os:.1e}", flush=True)
print(f"per forward (14 PVConvs): bf16x6 {tot_old:.0f} us, fp16x3 {tot_new:.0f} us, output-stationary {tot_os:.0f} us")
