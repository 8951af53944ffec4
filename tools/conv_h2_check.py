"""fp16x3 voxel convolution: time + SHA-1 of the outputs (values and GroupNorm slice partials) per denoiser shape, for A/B runs of
two builds / switches (the lines must be identical up to the times).  python tools/conv_h2_check.py [B]"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def sha(*ts):
    h = hashlib.sha1()
    for x in ts:
        h.update(x.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:12]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
for cin, cout, r, groups in [(64, 64, 32, 8), (32, 32, 32, 8), (128, 128, 16, 8), (64, 64, 16, 8), (256, 256, 8, 8), (512, 256, 8, 8), (128, 128, 8, 8),
                             (48, 40, 16, 0), (20, 72, 8, 0)]:
    x = torch.randn(B, cin, r ** 3).cuda()
    wt = (torch.randn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5).cuda()
    bias = torch.randn(cout).cuda()
    w3 = ops.conv3d_h2_pack(wt)
    xh = ops.to_h2(x)
    fl = 2 * 27 * cin * cout * r ** 3 * B
    y = ops.conv3d_h2(xh, w3, bias, cin, cout, r)
    line = f"{cin:4d}->{cout:4d} r={r:2d} B={B:2d}  y {sha(y)}"
    if groups:
        y2, st = ops.conv3d_h2_gn(xh, w3, bias, cin, cout, r, groups)
        part = st[0].view(torch.float64)[:B * groups * st[1] * 2]
        line += f"  gn {sha(y2, part)}"
        t3 = t(lambda: ops.conv3d_h2_gn(xh, w3, bias, cin, cout, r, groups))
    else:
        t3 = t(lambda: ops.conv3d_h2(xh, w3, bias, cin, cout, r))
    print(line + f"   | {t3:7.1f} us ({3 * fl / t3 / 1e9:5.2f} PF exec)")
