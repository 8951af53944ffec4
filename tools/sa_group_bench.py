"""Grouping gather (bdm_sa_group) at the denoisers' four set-abstraction levels, B shapes: the LDS-staged kernel (round 6: channel rows brought
in by LDS-DMA, gathered from LDS) against the point-major repack + row gather, with the ball query of the level beside it and the pair
priced against HBM by SURVEY.md 8d's byte formula.  usage: sa_group_bench.py [B=16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops
from bdm_amd import functional as F


def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(0)
pts = (torch.randn(B, 3, 4096, generator=g) * 0.5).cuda()
for lvl, (n, m, r, c) in enumerate([(4096, 1024, 0.1, 32), (1024, 256, 0.2, 64), (256, 64, 0.4, 128), (64, 16, 0.8, 256)]):
    ctr = F.furthest_point_sample(pts, m)
    f = torch.randn(B, c, n, generator=g).cuda()
    idx = F.ball_query(ctr, pts, r, 32)
    a = ops.sa_group(pts, ctr, f, idx, point_major=False)
    b_ = ops.sa_group(pts, ctr, f, idx, point_major=True)
    assert torch.equal(a, b_)
    tq = t(lambda: F.ball_query(ctr, pts, r, 32))
    td = t(lambda: ops.sa_group(pts, ctr, f, idx, point_major=False))
    tp = t(lambda: ops.sa_group(pts, ctr, f, idx, point_major=True))
    byt = 4.0 * B * (3 * n + 3 * m + c * n + m * 32 + (c + 3) * m * 32)
    best = min(td, tp)
    print(f"level {lvl}: {B} x {n} -> {m} x 32, {c} ch: ball query {tq:6.1f} us | gather from LDS rows {td:6.1f} us, point-major repack + gather {tp:6.1f} us | "
          f"pair {byt / 2 ** 20:6.1f} MB / {tq + best:6.1f} us = {byt / (tq + best) / 1e6:5.2f} TB/s = {byt / (tq + best) / 8e6:.3f} of 8 TB/s", flush=True)
    pts = ctr
