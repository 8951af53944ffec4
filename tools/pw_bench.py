"""bdm_pointwise_conv microbenchmark: back-to-back launches of one shape.  usage: pw_bench.py  (shapes inside)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdm_amd import ops

def t(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

SHAPES = [(16, 256, 256, 64), (16, 256, 256, 128), (16, 256, 64, 64), (16, 256, 1024, 64), (16, 32, 256, 64), (1, 256, 256, 64),
          (16, 256, 256, 256), (16, 256, 832, 64), (16, 1536, 512, 16), (16, 512, 512, 16), (16, 128, 579, 4096),
          (16, 64, 32, 32768), (16, 64, 64, 4096), (16, 128, 64, 8192), (16, 32, 35, 32768), (16, 256, 384, 1024), (16, 32, 390, 4096)]
for B, M, K, n in SHAPES:
    x = torch.randn(B, K, n, device="cuda")
    w = torch.randn(M, K, device="cuda") / K ** 0.5
    b = torch.zeros(M, device="cuda")
    out = torch.empty(B, M, n, device="cuda")
    us = t(lambda: ops.pointwise_conv(x, w, b, out=out))
    fl = 2.0 * B * M * K * n
    by = 4.0 * (B * K * n + M * K + B * M * n)
    print(f"B={B:2d} M={M:4d} K={K:4d} n={n:5d}  {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  {by / us / 1e3:7.1f} GB/s", flush=True)
