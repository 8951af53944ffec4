"""Where does a round of the furthest-point sampler go?  Debug build with -DFPS_TIMING (bdm_amd/libbdm_hip_fpstiming.so): shader-clock sums per
phase over rounds 64 .. 191 of every wave (each stamp waits for all outstanding memory operations first, so the phases are serialised:
attribution, not a speed measurement).   BDM_LIB_PATH=bdm_amd/libbdm_hip_fpstiming.so python tools/fps_phase_probe.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import _lib as L
from bdm_amd import functional as F
lib = L.lib()
lib.bdm_debug_fps_timestamps.argtypes = [ctypes.c_void_p]
names = ["centre fetch (LDS)", "distances + thread max", "thread argmax + key", "wave max + winner rank", "slot write + barrier", "slot read + fold", "rank -> index, output"]
for B, n, m in [(1, 1024, 1024), (16, 1024, 256), (16, 4096, 1024)]:
    g = torch.Generator().manual_seed(0)
    pts = (torch.randn(B, 3, n, generator=g) * 0.5).cuda()
    for _ in range(2):
        F.furthest_point_sample(pts, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); F.furthest_point_sample(pts, m); e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1) * 1e3
    buf = torch.zeros(B * 16 * 8, dtype=torch.int64, device="cuda")
    lib.bdm_debug_fps_timestamps(buf.data_ptr())
    e0.record(); F.furthest_point_sample(pts, m); e1.record(); torch.cuda.synchronize()
    lib.bdm_debug_fps_timestamps(None)
    t = buf.view(B, 16, 8).double().cpu() / 128.0           # cycles per round
    live = t[:, :, :7].sum(-1) > 0
    w0 = t[0, 0, :7]
    print(f"B={B} n={n} m={m}: {plain:.1f} us untimed (incl. gather) = {plain * 1e3 / (m - 1):.0f} ns / round; stamped run {e0.elapsed_time(e1) * 1e3:.1f} us; "
          f"waves {int(live[0].sum())}; wave 0 cycles per round: total {float(w0.sum()):.0f}")
    for i, nm in enumerate(names):
        allw = t[0][live[0]][:, i]
        print(f"    {nm:<28s} wave0 {float(w0[i]):7.1f}   mean over waves {float(allw.mean()):7.1f}   max {float(allw.max()):7.1f}")
