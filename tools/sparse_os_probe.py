"""Timing probe of the output-stationary sparse first convolution (sparse_conv_os.hip) per layer shape; BDM_OS_DBG selects a debug mode of
the kernel (1: skip every MFMA group, 2: skip none, 3: prologue + epilogue only).  usage: sparse_os_probe.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops
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
for m in (1024, 256, 64):
    clouds[m] = F.furthest_point_sample(clouds[m * 4], m)
LAYERS = [("SA0.1", 32, 32, 32, 4096), ("SA1.0", 128, 64, 16, 1024), ("SA2.0", 192, 128, 8, 256),
          ("FP0.x", 256, 256, 8, 64), ("FP1.x", 256, 256, 8, 256), ("FP2.x", 128, 128, 16, 1024), ("FP3.x", 64, 64, 32, 4096)]
from bdm_amd import _lib as L
lib = L.lib()
for name, cin, cout, r, n in LAYERS:
    pts = clouds[n]
    ops.clear_plan_cache()
    plan = ops.voxel_plan(pts, r)
    f = torch.randn(B, cin, n, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    bias = torch.zeros(cout).cuda()
    packed_w, inv_scale = ops.conv3d_h2_pack(w)
    xr = torch.empty(B, (cin + 7) // 8, plan.n_max, 8, dtype=torch.float32, device="cuda")
    amax = torch.zeros(B, device="cuda")
    feat = lambda: L.check(lib.bdm_sparse_voxel_features_f32(B, cin, n, r, plan.n_max, L.ptr(f), cin * n, n, L.ptr(plan.cnt), L.ptr(plan.ws),
                                                            L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(xr), L.ptr(amax), L.stream()))
    feat()
    out = torch.empty(B, cout, r ** 3, device="cuda")
    conv = lambda: L.check(lib.bdm_sparse_conv_os(B, cin, cout, r, plan.n_max, L.ptr(xr), L.ptr(amax), L.ptr(plan.occ_index), L.ptr(packed_w),
                                                 L.ptr(inv_scale), L.ptr(bias), L.ptr(out), L.stream()))
    dil = lambda: L.check(lib.bdm_sparse_conv_dil(B, cin, cout, r, plan.n_max, plan.n_dil_max, L.ptr(xr), L.ptr(amax), L.ptr(plan.occ_index),
                                                 L.ptr(plan.dil_list), L.ptr(plan.tile_start), L.ptr(plan.plane_start), L.ptr(packed_w),
                                                 L.ptr(inv_scale), L.ptr(bias), L.ptr(out), L.stream()))
    ntl = plan.tile_start[:, -1].float().mean().item()
    tdil = t(lambda: L.check(lib.bdm_voxel_dilate(B, r, plan.n_dil_max, L.ptr(plan.cnt), L.ptr(plan.dil_list), L.ptr(plan.plane_start), L.ptr(plan.tile_start), L.stream())))
    row = [f"{name} {cin:4d}->{cout:4d} r={r:2d} n_occ={float(plan.n_occ.float().mean()):7.1f} features {t(feat):6.1f} us | dilate {tdil:5.1f} us, "
           f"{ntl:4.1f} tiles/shape, compact conv {t(dil):6.1f} us | brick conv"]
    for mode in ("0", "1", "2", "3", "4", "5"):
        os.environ["BDM_OS_DBG"] = mode
        row.append(f"dbg{mode} {t(conv):6.1f}")
    os.environ["BDM_OS_DBG"] = "0"
    print("  ".join(row), flush=True)
    # activity of the (16-voxel block, tap quad) fragments on THIS plan, computed on the host from occ_index
    occ = (plan.occ_index.view(B, r, r, r) >= 0).cpu().numpy()
    import numpy as np
    zb = min(16, r)
    acts = []
    for b in range(min(B, 4)):
        P = np.zeros((r + 2,) * 3, bool); P[1:-1, 1:-1, 1:-1] = occ[b]
        act = np.zeros((r, r, r // zb, 27), bool)
        for tp in range(27):
            dx, dy, dz = tp // 9 - 1, (tp // 3) % 3 - 1, tp % 3 - 1
            act[..., tp] = P[1 + dx:1 + dx + r, 1 + dy:1 + dy + r, 1 + dz:1 + dz + r].reshape(r, r, r // zb, zb).any(-1)
        acts.append(np.stack([act[..., 4 * Q:min(4 * Q + 4, 27)].any(-1) for Q in range(7)], -1).mean())
    print(f"      host-side fragment activity {100 * np.mean(acts):.1f} %  occupied {100 * occ.mean():.1f} %")
