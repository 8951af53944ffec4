"""Timing probe of the output-stationary sparse first convolution (sparse_conv_os.hip) per layer shape at the bench's batch: feature records,
dilated plan, the convolution in its compact and dense output forms, and the operand split that consumes each.  usage: sparse_os_probe.py [B]"""
import os, sys, torch
import torch.nn as nn
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
for m in (1024, 256, 64):
    clouds[m] = F.furthest_point_sample(clouds[m * 4], m)
LAYERS = [("SA0.1", 32, 32, 32, 4096), ("SA1.0", 128, 64, 16, 1024), ("SA2.0", 192, 128, 8, 256),
          ("FP0.x", 256, 256, 8, 64), ("FP1.x", 256, 256, 8, 256), ("FP2.x", 128, 128, 16, 1024), ("FP3.x", 64, 64, 32, 4096)]
lib = L.lib()
for name, cin, cout, r, n in LAYERS:
    pts = clouds[n]
    ops.clear_plan_cache()
    plan = ops.voxel_plan(pts, r)
    f = torch.randn(B, cin, n, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    bias = torch.zeros(cout).cuda()
    gn = nn.GroupNorm(8, cout).cuda()
    pk = ops.conv3d_h2_pack(w)
    xr = torch.empty(B, (cin + 7) // 8, plan.n_max, 8, dtype=torch.float32, device="cuda")
    amax = torch.zeros(B, device="cuda")
    feat = lambda: L.check(lib.bdm_sparse_voxel_features_f32(B, cin, n, r, plan.n_max, L.ptr(f), cin * n, n, L.ptr(plan.cnt), L.ptr(plan.ws),
                                                            L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(xr), L.ptr(amax), L.stream()))
    feat()
    yd = torch.empty(B, cout, r ** 3, device="cuda")
    yc = torch.empty(B, plan.n_dil_max, cout, device="cuda")
    part = torch.empty(B, 8, plan.tile_start.shape[1], 2, dtype=torch.float64, device="cuda")
    import ctypes
    sl = ctypes.c_int(0)

    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")

    def conv(y, compact):
        return lambda: (ctr.zero_(), L.check(lib.bdm_sparse_conv_dil_gn(B, cin, cout, r, plan.n_max, plan.n_dil_max, L.ptr(xr), L.ptr(amax), L.ptr(plan.occ_index),
                                                          L.ptr(plan.dil_list), L.ptr(plan.dil_index), L.ptr(plan.tile_start),
                                                          L.ptr(pk[0]), L.ptr(pk[1]), L.ptr(bias), L.ptr(y), compact, 8, L.ptr(part), ctypes.byref(sl), L.ptr(ctr), L.stream())))
    tdil = t(lambda: L.check(lib.bdm_voxel_dilate(B, r, plan.n_dil_max, L.ptr(plan.cnt), L.ptr(plan.dil_list), L.ptr(plan.dil_index), L.ptr(plan.plane_start),
                                                  L.ptr(plan.tile_start), L.stream())))
    ntl = plan.tile_start[:, 0, 7].float().mean().item()
    nd = plan.tile_start[:, :, 1].max(1).values.float().mean().item()
    tc, td = t(conv(yc, 1)), t(conv(yd, 0))
    st = (part, part.shape[2], 8)
    comp = ops.CompactGrid(yc, plan, bias, cout)
    th_c = t(lambda: ops.to_h2(comp, gn, swish=True, stats=st))
    th_d = t(lambda: ops.to_h2(yd, gn, swish=True, stats=st))
    print(f"{name} {cin:4d}->{cout:4d} r={r:2d} n_occ={float(plan.n_occ.float().mean()):7.1f} n_dil={nd:7.1f} ({ntl:4.1f} tiles) | features {t(feat):6.1f} us, "
          f"dilate {tdil:5.1f} us | conv compact {tc:6.1f} us + split {th_c:5.1f} us | conv dense {td:6.1f} us + split {th_d:5.1f} us", flush=True)
