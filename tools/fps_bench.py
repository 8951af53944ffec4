"""FPS timing (B=16): python tools/fps_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import functional as F
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n, m in [(4096, 1024), (1024, 256), (256, 64), (64, 16), (8192, 1024), (16384, 1024)]:
    pts = torch.randn(16, 3, n).cuda()
    us = t(lambda: F.furthest_point_sample(pts, m))
    print(f"N={n} M={m}: {us:8.1f} us ({us / (m - 1) * 1e3:5.0f} ns/round)")
