"""1x1 GEMM layer shapes of a B = 16 forward, plain and with statistics / folded input.  A/B: BDM_LIB_PATH=<other build> python tools/pw_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn
from bdm_amd import ops
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = [0.0, 0.0, 0.0]
for B, M, K, n in [(16, 128, 192, 4096), (16, 128, 64, 4096), (16, 128, 128, 4096), (16, 64, 96, 4096), (16, 128, 64, 8192), (16, 64, 67, 8192), (16, 128, 131, 2048),
                   (16, 256, 128, 2048), (16, 128, 256, 1024), (16, 128, 128, 1024), (16, 256, 384, 1024), (16, 128, 579, 4096), (16, 64, 32, 32768)]:
    x = torch.randn(B, K, n, device="cuda"); w = torch.randn(M, K, device="cuda") / K ** 0.5; b = torch.zeros(M, device="cuda")
    out = torch.empty(B, M, n, device="cuda")
    a = t(lambda: ops.pointwise_conv(x, w, b, out=out))
    s = t(lambda: ops.pointwise_conv_gn(x, w, b, out=out, out_groups=8))
    f = float("nan")
    if ops.gn_foldable(K, 8):
        gn = nn.GroupNorm(8, K).cuda()
        _, st = ops.pointwise_conv_gn(x, torch.eye(K, device="cuda"), None, out_groups=8)
        f = t(lambda: ops.pointwise_conv_gn(x, w, b, out=out, fold_in=(st, gn), out_groups=8))
    tot[0] += a; tot[1] += s; tot[2] += 0 if f != f else f
    print(f"M={M:4d} K={K:4d} n={n:6d}: plain {a:6.1f}  +stats {s:6.1f}  fold+stats {f:6.1f} us", flush=True)
print(f"sum: plain {tot[0]:.1f}  +stats {tot[1]:.1f}  fold+stats {tot[2]:.1f} us")
