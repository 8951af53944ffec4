import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops
from bdm_amd import functional as F

def t(fn, n=20):
    """GPU time per call: n calls captured into one HIP graph (no host launch gaps), replayed 5 times."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3

B, U = 16, 32
CASES = [(4096, 1024, 0.1, 32, 0.25), (4096, 1024, 0.1, 32, 0.05), (1024, 256, 0.2, 64, 0.25), (16384, 1024, 0.1, 32, 0.25)]
for n, m, r, c, scale in ([CASES[int(sys.argv[1])]] if len(sys.argv) > 1 else CASES):
    pts = (torch.randn(B, 3, n) * scale).cuda()
    ctr = F.furthest_point_sample(pts, m)
    f = torch.randn(B, c, n).cuda()
    bytes_alg = 4 * (3 * n + 3 * m + c * n + m * U + (c + 3) * m * U) * B
    t_bq = t(lambda: F.ball_query(ctr, pts, r, U))
    idx = F.ball_query(ctr, pts, r, U)
    t_g0 = t(lambda: ops.sa_group(pts, ctr, f, idx, point_major=False))
    t_gr = t(lambda: ops.sa_group(pts, ctr, f, idx))
    t_fu = t_bq + t_gr
    filled = float((idx != idx[:, :, :1]).any(-1).float().mean())
    print(f"N={n} M={m} r={r} C={c} scale={scale}: ball_query {t_bq:6.1f} us  sa_group {t_gr:6.1f} us (direct {t_g0:6.1f})  sum {t_fu:6.1f} us  "
          f"alg {bytes_alg/1e6:6.1f} MB -> {bytes_alg/t_fu/1e6:5.2f} TB/s ({bytes_alg/t_fu/1e6/8*100:4.1f}% of 8 TB/s); centres with >1 hit {filled:.2f}")
