import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops

def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = 16
for cin, cout, r, npts in [(64, 64, 32, 4096), (390, 32, 32, 4096), (32, 32, 32, 4096), (128, 128, 16, 1024), (256, 256, 8, 256), (256, 256, 8, 64)]:
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(B, 3, npts, generator=g).cuda()
    nc, vc = ops.voxel_coords(pts, r)
    f = torch.randn(B, cin, npts, generator=g).cuda()
    vox, rowocc = ops.avg_voxelize(f, vc, r, with_row_occupancy=True)
    w = ops.conv3d_pack((torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda())
    bias = torch.zeros(cout).cuda()
    fl = 2 * 27 * cin * cout * r ** 3 * B
    td = t(lambda: ops.conv3d(vox, w, bias, r))
    ts = t(lambda: ops.conv3d(vox, w, bias, r, rowocc=rowocc))
    tz = t(lambda: ops.conv3d(vox, w, bias, r, rowocc=torch.zeros_like(rowocc)))
    to = t(lambda: ops.conv3d(vox, w, bias, r, rowocc=torch.ones_like(rowocc)))
    print(f"{cin:4d}->{cout:4d} r={r:2d} occ_rows={float(rowocc.float().mean()):.2f} occ_vox={float((vox[:,0]!=0).float().mean()):.3f} "
          f"dense {td:7.1f} us ({fl/td/1e6:6.1f} TF/s)  sparse {ts:7.1f} us  all-empty {tz:7.1f} us  all-full(sparse path) {to:7.1f} us")

print("--- synthetic occupancy patterns, 64->64 r=32")
cin, cout, r = 64, 64, 32
vox = torch.randn(B, cin, r ** 3).cuda()
w = ops.conv3d_pack((torch.randn(cout, cin, 3, 3, 3) / 40).cuda())
bias = torch.zeros(cout).cuda()
xs, ys = torch.meshgrid(torch.arange(r), torch.arange(r), indexing="ij")
pats = {"slab x<8": xs < 8, "slab y<8": ys < 8, "disc r=11": (xs - 15.5) ** 2 + (ys - 15.5) ** 2 < 121,
        "random 36%": torch.rand(r, r) < 0.36, "one row": (xs == 5) & (ys == 5), "full": xs >= 0}
for name, m in pats.items():
    ro = m.reshape(1, -1).expand(B, -1).contiguous().to(torch.uint8).cuda()
    print(f"{name:12s} frac={float(m.float().mean()):.2f}  {t(lambda: ops.conv3d(vox, w, bias, r, rowocc=ro)):8.1f} us")

print("--- bf16x6 vs fp32 MFMA")
for cin, cout, r in [(64, 64, 32), (390, 32, 32), (32, 32, 32), (128, 128, 16), (128, 64, 16), (256, 256, 8)]:
    x = torch.randn(B, cin, r ** 3).cuda()
    wt = (torch.randn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5).cuda()
    bias = torch.zeros(cout).cuda()
    w32, w16 = ops.conv3d_pack(wt), ops.conv3d_s3_pack(wt)
    xs = ops.to_s3(x)
    fl = 2 * 27 * cin * cout * r ** 3 * B
    t32 = t(lambda: ops.conv3d(x, w32, bias, r))
    t16 = t(lambda: ops.conv3d_s3(xs, w16, bias, cin, cout, r))
    tcv = t(lambda: ops.to_s3(x))
    print(f"{cin:4d}->{cout:4d} r={r:2d}  fp32 {t32:7.1f} us ({fl/t32/1e6:6.1f} TF/s)   bf16x6 {t16:7.1f} us ({fl/t16/1e6:6.1f} TF/s)   to_s3 {tcv:6.1f} us")
