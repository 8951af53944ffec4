"""Timing probe of the list convolutions (sparse_conv_os.hip) per layer shape and TILE FORM at the bench's batch: first convolution on the
once-dilated list (compact rows + GroupNorm partials) and second convolution on the twice-dilated list, full tiles (BDM_DIL_TILE=0: one
workgroup per CU) against half tiles of 256 / 128 / 64 entries (two per CU).  usage: sparse_os_probe.py [B] [spread]"""
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
N0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4096            # points of the cloud (C4: 8192)
g = torch.Generator().manual_seed(0)
clouds = {N0: (torch.randn(B, 3, N0, generator=g) * 0.5).cuda()}
for m in (N0 // 4, N0 // 16, N0 // 64):
    clouds[m] = F.furthest_point_sample(clouds[m * 4], m)
LAYERS = [("SA0.1", 32, 32, 32, N0), ("FP3.x", 64, 64, 32, N0), ("FP2.x", 128, 128, 16, N0 // 4), ("SA1.0", 128, 64, 16, N0 // 4)]
ONLY_LIST = len(sys.argv) > 3
for name, cin, cout, r, n in LAYERS:
    pts = clouds[n]
    f = torch.randn(B, cin, n, generator=g).cuda()
    w1 = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    w2 = (torch.randn(cout, cout, 3, 3, 3, generator=g) / (27 * cout) ** 0.5).cuda()
    b1, b2 = torch.randn(cout, generator=g).cuda(), torch.randn(cout, generator=g).cuda()
    gn1 = nn.GroupNorm(8, cout).cuda()
    pk1, pk2, wsum = ops.conv3d_h2_pack(w1), ops.conv3d_h2_pack(w2), ops.conv_class_pack(w2)
    base = None
    for form in ("0", "256", "128", "64"):
        ops.DIL_TILE = form
        ops.clear_plan_cache()
        plan = ops.voxel_plan(pts, r, dilate=2)
        first = lambda: ops.sparse_first_conv_os(f, plan, pk1, b1, cout, gn_groups=8, compact=True)
        comp, st1 = first()
        rows_h2, const_h2, const_f32, inv_s = ops.to_h2_rows(comp, plan, gn1, st1)
        second = lambda: ops.second_conv_rows(rows_h2, const_h2, const_f32, inv_s, plan, pk2, wsum, b2, cout, cout, 8)
        rows, _, _ = second()
        nd1 = plan.tile_start[:, -1, 1].float().mean().item(); nd2 = plan.d2_tiles[:, -1, 1].float().mean().item()
        nt1 = plan.tile_start[:, 0, 7].float().mean().item(); nt2 = plan.d2_tiles[:, 0, 7].float().mean().item()
        rmax1 = (plan.tile_start[:, :, 5] + plan.tile_start[:, :, 9] + plan.tile_start[:, :, 11]).max().item()
        rmax2 = (plan.d2_tiles[:, :, 5] + plan.d2_tiles[:, :, 9] + plan.d2_tiles[:, :, 11]).max().item()
        same = ""
        if base is None:
            base = (comp.rows.clone(), rows.clone())
        else:
            k1, k2 = int(nd1 * 0.5), int(nd2 * 0.5)
            same = f" bits==full: {bool(torch.equal(comp.rows[:, :k1], base[0][:, :k1]))}/{bool(torch.equal(rows[:, :k2], base[1][:, :k2]))}"
        tp1 = t(lambda: ops.plan_dilation.__wrapped__(plan) if hasattr(ops.plan_dilation, '__wrapped__') else None, 1)
        print(f"{name} {cin:3d}->{cout:3d} r={r:2d} tile {form:>3s}: listed {nd1:6.0f} / {nd2:6.0f} voxels, {nt1:5.1f} / {nt2:5.1f} tiles per shape, max rows {rmax1} / {rmax2} | "
              f"first {t(first):6.1f} us  second {t(second):6.1f} us{same}", flush=True)

if ONLY_LIST:
    sys.exit(0)
# ---- 8^3 levels: the list kernel in its half-tile form against the route the product takes there (fp16x3 GEMM over the occupied rows +
# gather with GroupNorm-1 + Swish + operand split in its epilogue)
print("8^3 levels: first convolution -> second convolution's fp16 operand", flush=True)
for name, cin, cout, n in [("FP1.x", 256, 256, N0 // 16), ("FP0.x", 256, 256, N0 // 64), ("SA2.0", 192, 128, N0 // 16), ("SA3->", 128, 256, N0 // 64)]:
    r = 8
    pts = clouds[n]
    f = torch.randn(B, cin, n, generator=g).cuda()
    w1 = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    b1 = torch.randn(cout, generator=g).cuda()
    gn1 = nn.GroupNorm(8, cout).cuda()
    pk1, pkg = ops.conv3d_h2_pack(w1), ops.sparse_conv_pack_h2(w1)
    ops.DIL_TILE = "0"
    ops.clear_plan_cache()
    plan = ops.voxel_plan(pts, r)
    def gemm():
        v, st = ops.sparse_first_conv_planned(f, plan, pkg, b1, cout, gn_groups=8)
        return ops.to_h2(v, gn1, swish=True, stats=st)
    ref = gemm()
    t_gemm = t(gemm)
    line = f"{name} {cin:3d}->{cout:3d} n={n:4d}: features + split + GEMM + gather(stats) + operand split {t_gemm:6.1f} us |"
    for form in ("0", "64", "128"):
        ops.DIL_TILE = form
        ops.clear_plan_cache()
        plan = ops.voxel_plan(pts, r, dilate=1)

        def lst():
            v, st = ops.sparse_first_conv_os(f, plan, pk1, b1, cout, gn_groups=8, compact=False)
            return ops.to_h2(v, gn1, swish=True, stats=st)
        got = lst()
        conv_only = t(lambda: ops.sparse_first_conv_os(f, plan, pk1, b1, cout, gn_groups=8, compact=False))
        d = float((got[0].float() - ref[0].float()).abs().max())
        line += f" list tile {form:>3s}: {t(lst):6.1f} us (conv + features {conv_only:6.1f}) max|d hi| {d:.1e} |"
    print(line, flush=True)
