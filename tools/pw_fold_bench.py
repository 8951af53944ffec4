"""What the GroupNorm folding costs the 1x1 GEMM: plain / + output statistics / + input fold / both, on the SharedMLP shapes of the
set-abstraction levels (B = 16).  python tools/pw_fold_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from bdm_amd import ops

def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B, M, K, n in [(16, 64, 32, 32768), (16, 32, 35, 32768), (16, 128, 64, 8192), (16, 64, 67, 8192), (16, 256, 128, 2048), (16, 128, 128, 4096), (16, 64, 128, 4096)]:
    x = torch.randn(B, K, n, device="cuda")
    w = torch.randn(M, K, device="cuda") / K ** 0.5
    b = torch.zeros(M, device="cuda")
    out = torch.empty(B, M, n, device="cuda")
    gn_in = nn.GroupNorm(8, K).cuda() if K % 8 == 0 else None
    plain = t(lambda: ops.pointwise_conv(x, w, b, out=out))
    stats = t(lambda: ops.pointwise_conv_gn(x, w, b, out=out, out_groups=8))
    line = f"B={B} M={M:4d} K={K:4d} n={n:6d}  plain {plain:6.1f} us | +stats {stats:6.1f}"
    if gn_in is not None:
        # statistics of x as a previous call would have left them
        w0 = torch.eye(K, device="cuda")
        _, st = ops.pointwise_conv_gn(x, w0, None, out_groups=8)
        fold = t(lambda: ops.pointwise_conv_gn(x, w, b, out=out, fold_in=(st, gn_in)))
        both = t(lambda: ops.pointwise_conv_gn(x, w, b, out=out, fold_in=(st, gn_in), out_groups=8))
        line += f" | +fold {fold:6.1f} | both {both:6.1f}"
    by = 4.0 * (B * K * n + B * M * n)
    print(line + f"   ({by / plain / 1e3:6.0f} GB/s plain)", flush=True)
