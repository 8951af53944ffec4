"""What would the SECOND convolution of a PVConv cost on the twice-dilated voxel set?  Emulation with the existing compact kernel: the
once-dilated set D1 plays the occupied set (rows = D1 entries), bdm_voxel_dilate on it gives D2 and its tiles, and
bdm_sparse_conv_dil runs over them with random rows.  Compared with the dense fp16x3 convolution at the same layer.  usage: [B]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops, _lib as L
from bdm_amd import functional as F


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(0)
clouds = {4096: (torch.randn(B, 3, 4096, generator=g) * 0.5).cuda()}
clouds[1024] = F.furthest_point_sample(clouds[4096], 1024)
lib = L.lib()
for name, c, r, n in [("32ch @32^3", 32, 32, 4096), ("64ch @32^3", 64, 32, 4096), ("64ch @16^3", 64, 16, 1024), ("128ch @16^3", 128, 16, 1024)]:
    ops.clear_plan_cache()
    plan = ops.plan_dilation(ops.voxel_plan(clouds[n], r))
    r3 = r ** 3
    cnt2 = (plan.dil_index >= 0).to(torch.int32).contiguous()
    tiles = lib.bdm_voxel_dilate_slices(r)
    dl2 = torch.empty(B, r3, dtype=torch.int32, device="cuda"); di2 = torch.empty(B, r3, dtype=torch.int32, device="cuda")
    ps2 = torch.empty(B, r + 2, dtype=torch.int32, device="cuda"); ts2 = torch.empty(B, tiles, 8, dtype=torch.int32, device="cuda")
    L.check(lib.bdm_voxel_dilate(B, r, r3, L.ptr(cnt2), L.ptr(dl2), L.ptr(di2), L.ptr(ps2), L.ptr(ts2), L.stream()))
    n1 = plan.tile_start[:, :, 1].max(1).values.float().mean().item()
    n2 = ts2[:, :, 1].max(1).values.float().mean().item()
    w = (torch.randn(c, c, 3, 3, 3, generator=g) / (27 * c) ** 0.5).cuda()
    bias = torch.zeros(c).cuda()
    pk = ops.conv3d_h2_pack(w)
    xr = torch.randn(B, (c + 7) // 8, r3, 8, generator=g).cuda()      # rows indexed by D1 rank
    amax = torch.full((B,), 4.0, device="cuda")
    y = torch.empty(B, r3, c, device="cuda")
    part = torch.empty(B, 8, tiles, 2, dtype=torch.float64, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    sl = ctypes.c_int(0)
    conv = lambda: (ctr.zero_(), L.check(lib.bdm_sparse_conv_dil_gn(B, c, c, r, r3, r3, L.ptr(xr), L.ptr(amax), L.ptr(plan.dil_index), L.ptr(dl2), L.ptr(di2),
                                                                  L.ptr(ts2), L.ptr(pk[0]), L.ptr(pk[1]), L.ptr(bias), L.ptr(y), 1, 8, L.ptr(part),
                                                                  ctypes.byref(sl), L.ptr(ctr), L.stream())))
    xd = torch.randn(B, c, r3, generator=g).cuda()
    xh = ops.to_h2(xd, scale=1024.0)
    dense = lambda: ops.conv3d_h2_gn(xh, pk, bias, c, c, r, 8)
    print(f"{name}: once-dilated {n1:7.0f} voxels / shape, twice-dilated {n2:7.0f} ({100 * n2 / r3:4.1f} % of the grid, {ts2[:, 0, 7].float().mean().item():4.1f} tiles) | "
          f"compact kernel over D2 {t(conv):6.1f} us | dense convolution {t(dense):6.1f} us", flush=True)
