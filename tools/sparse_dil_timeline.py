"""Phase timeline of sconv_dil_kernel's workgroups (debug build with -DDIL_TIMING: bdm_amd/libbdm_hip_diltiming.so; wall_clock64 stamps).
usage: BDM_LIB_PATH=bdm_amd/libbdm_hip_diltiming.so python tools/sparse_dil_timeline.py"""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops, _lib as L
from bdm_amd import functional as F
B = 16
g = torch.Generator().manual_seed(0)
clouds = {4096: (torch.randn(B, 3, 4096, generator=g) * 0.5).cuda()}
for m in (1024, 256, 64):
    clouds[m] = F.furthest_point_sample(clouds[m * 4], m)
lib = L.lib()
lib.bdm_debug_dil_timestamps.argtypes = [ctypes.c_void_p]
FORMS = sys.argv[1:] or ["0", "256"]
for form, (name, cin, cout, r, n) in [(fm, l) for l in [("SA0.1", 32, 32, 32, 4096), ("FP2.x", 128, 128, 16, 1024), ("FP3.x", 64, 64, 32, 4096)] for fm in FORMS]:
    ops.DIL_TILE = form
    ops.clear_plan_cache()
    plan = ops.voxel_plan(clouds[n], r, dilate=1)
    f = torch.randn(B, cin, n, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    bias = torch.zeros(cout).cuda()
    pk = ops.conv3d_h2_pack(w)
    tiles = plan.tile_start.shape[1]
    nblk = B * ((cout + 63) // 64 if cout > 32 else 1) * tiles
    ts = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.sparse_first_conv_os(f, plan, pk, bias, cout, gn_groups=8, compact=True)
    torch.cuda.synchronize()
    lib.bdm_debug_dil_timestamps(ts.data_ptr())
    ops.sparse_first_conv_os(f, plan, pk, bias, cout, gn_groups=8, compact=True)
    torch.cuda.synchronize()
    lib.bdm_debug_dil_timestamps(None)
    t = ts.cpu().numpy().reshape(-1, 8).astype(np.float64)
    live = t[t[:, 6] > 0]
    t0 = live[:, 0].min()
    us = (live - t0) / 100.0          # 100 MHz
    d = np.diff(us[:, :7], axis=1)
    lab = ["tile record -> neighbour records", "-> first barrier (W/X chunk 0 loaded)", "stage chunk 0", "chunk 0 MFMA (+ prefetch)", "chunks 1..", "epilogue"]
    hw = live[:, 7].astype(np.int64)
    cu = ((hw >> 32) & 15) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 15)     # (XCC, SE, SH, CU)
    conc = 0
    for c in np.unique(cu):
        iv = us[cu == c][:, [0, 6]]
        ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
        k = 0
        for _, dlt in ev:
            k += dlt
            conc = max(conc, k)
    print(f"tile form {form}: {len(np.unique(cu))} distinct CUs, at most {conc} workgroups resident on one CU at a time")
    print(f"{name} {cin}->{cout} r={r}: {len(live)} live workgroups; start spread {us[:, 0].max():.1f} us; kernel end {us[:, 6].max():.1f} us")
    for i, l in enumerate(lab):
        print(f"    {l:42s} mean {d[:, i].mean():7.2f} us   max {d[:, i].max():7.2f}")
    st = np.sort(us[:, 0])
    print("    start times (us), deciles:", " ".join(f"{st[int(q * (len(st) - 1))]:.1f}" for q in np.linspace(0, 1, 11)))
    print(f"    workgroup lifetime                         mean {(us[:, 6] - us[:, 0]).mean():7.2f} us   max {(us[:, 6] - us[:, 0]).max():7.2f}")
