"""fp16x3 vs bf16x6 voxel convolution on the shapes of the denoisers (B = 16). python tools/conv_h2_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S3 = os.environ.get("BENCH_S3", "1") == "1"
for cin, cout, r in [(64, 64, 32), (32, 32, 32), (128, 128, 16), (64, 64, 16), (256, 256, 8), (512, 256, 8), (128, 128, 8)]:
    x = torch.randn(B, cin, r ** 3).cuda()
    wt = (torch.randn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5).cuda()
    bias = torch.zeros(cout).cuda()
    w6, w3 = ops.conv3d_s3_pack(wt), ops.conv3d_h2_pack(wt)
    xs, xh = ops.to_s3(x), ops.to_h2(x)
    fl = 2 * 27 * cin * cout * r ** 3 * B
    t6 = t(lambda: ops.conv3d_s3(xs, w6, bias, cin, cout, r)) if S3 else float('nan')
    t3 = t(lambda: ops.conv3d_h2(xh, w3, bias, cin, cout, r))
    print(f"{cin:4d}->{cout:4d} r={r:2d}  bf16x6 {t6:7.1f} us ({fl/t6/1e6:6.1f} TF/s eq, {6*fl/t6/1e9:5.2f} PF exec)   "
          f"fp16x3 {t3:7.1f} us ({fl/t3/1e6:6.1f} TF/s eq, {3*fl/t3/1e9:5.2f} PF exec)")
