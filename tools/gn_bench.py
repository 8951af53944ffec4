"""GroupNorm(8)+Swish kernel timing on the denoiser's tensor shapes (B=16), inside a HIP graph (no host gaps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
B = 16
for C, L in [(64, 4096), (128, 4096), (32, 4096), (256, 512), (256, 64), (512, 16), (128, 1024), (64, 32768), (32, 32768), (128, 8192)]:
    x = torch.randn(B, C, L).cuda(); ga = torch.ones(C).cuda(); be = torch.zeros(C).cuda()
    us = t(lambda: ops.group_norm_(x, ga, be, 8, 1e-5, swish=True))
    mb = 2 * x.numel() * 4 / 1e6
    print(f"C={C:4d} L={L:6d} chunk={C // 8 * L:7d} floats: {us:7.1f} us  ({mb:7.1f} MB r+w -> {mb / us:5.2f} TB/s)")
