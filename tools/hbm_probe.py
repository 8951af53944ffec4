"""Practical streaming rates of this box with stock kernels (the yardstick for the HBM-bound classes): read-only reduction, copy, in-place
scale, for a cache-sized (128 MB) and an HBM-sized (2 GB) tensor.  python tools/hbm_probe.py"""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (128, 2048):
    n = mb * 2 ** 20 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    s = t(lambda: x.sum()); c = t(lambda: y.copy_(x)); m = t(lambda: x.mul_(1.0001))
    print(f"{mb:5d} MB: sum {4 * n / s / 1e12:5.2f} TB/s read | copy {8 * n / c / 1e12:5.2f} TB/s r+w | mul_ {8 * n / m / 1e12:5.2f} TB/s r+w")
