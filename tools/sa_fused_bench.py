"""First set-abstraction level's grouped MLP: the operator chain (sa_group + two folded 1x1 GEMMs + folded max) against the three
recompute passes of bdm_sa_mlp2_fused, per batch.  python tools/sa_fused_bench.py [B n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdm_amd.modules import PointNetSAModule

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

CASES = [(16, 4096), (8, 4096), (1, 4096), (16, 16384)]
if len(sys.argv) > 2:
    CASES = [(int(sys.argv[1]), int(sys.argv[2]))]
for B, n in CASES:
    torch.manual_seed(0)
    sa = PointNetSAModule(1024, 0.1, 32, in_channels=32, out_channels=[32, 64]).cuda()
    coords = torch.rand(B, 3, n, device="cuda") - 0.5
    feats = torch.randn(B, 32, n, device="cuda")
    temb = torch.randn(B, 64, device="cuda")[:, :, None].expand(-1, -1, n)
    with torch.no_grad():
        centers, idx = sa.plan(coords)
        ev = torch.cuda.Event(); ev.record()
        def run(fused):
            sa.fuse_mlp = fused
            sa._planned = (centers, idx, ev, coords)
            return sa((feats, coords, temb))[0]
        a, b = run(True), run(False)
        err = float((a - b).norm() / b.norm())
        print(f"B={B:3d} n={n:6d}: chain {t(lambda: run(False)):7.1f} us   fused {t(lambda: run(True)):7.1f} us   rel diff {err:.2e}", flush=True)
