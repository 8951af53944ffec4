"""Error budget of one denoiser forward: how far is the HIP path, and how far is the fp32 CPU oracle, from the float64 evaluation of the
same network on the same fp32-determined geometry (oracle/truth.py)?  Their mutual distance (what the parity tests bound) is the sum of
the two; the full-length trajectory amplifies it ~500x, so this is where the 1e-3 margin is made or lost.

    python tools/error_budget.py [pc2|pvd] [N] [B]

Prints one line per kernel-form variant of the HIP path (default forms at this batch, the forms a B = 16 batch picks, bf16x6
convolutions, unsplit time embedding, ...): rel-L2 of the predicted noise vs truth64 and vs oracle32.
"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from helpers import point_cloud_inputs, rel_l2  # noqa: E402
from bdm_amd import ops, pvcnn  # noqa: E402
from bdm_amd.modules import PVConv, PointNetSAModule, SharedMLP  # noqa: E402
from bdm_amd.pvcnn import PVCNN2_PC2, PVCNN2_PVD  # noqa: E402
from bdm_amd.utils.procedural import fill_module_  # noqa: E402
from oracle import ops as O, ref_net, truth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "pc2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
O.build()
torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
extra = 387 if which == "pc2" else 0
net = fill_module_((PVCNN2_PC2 if which == "pc2" else PVCNN2_PVD)(3, 64, extra_feature_channels=extra).eval(), seed=11)
x = point_cloud_inputs(B, 3 + extra, N, seed=4200 + N)
t = (torch.arange(B) * 311 + 730) % 1000
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
t0 = time.time()
ref32 = ref_net.pvcnn_forward(sd, x, t)
t1 = time.time()
ref64 = truth.pvcnn_forward_f64(sd, x, t)
print(f"{which} N={N} B={B}: oracle32 {t1 - t0:.1f} s, truth64 {time.time() - t1:.1f} s; oracle32 vs truth64 {rel_l2(ref32.double(), ref64):.3e}", flush=True)
if not torch.cuda.is_available():
    sys.exit(0)
net = net.cuda()
xd, td = x.cuda(), t.cuda()


def run(label, **patch):
    saved = []
    for key, val in patch.items():
        owner, attr = {"pv": PVConv, "ops": ops, "pvcnn": pvcnn, "sa": PointNetSAModule, "mlp": SharedMLP}[key.split("__")[0]], key.split("__")[1]
        saved.append((owner, attr, getattr(owner, attr)))
        setattr(owner, attr, val)
    try:
        y = net(xd, td).cpu()
    finally:
        for owner, attr, val in saved:
            setattr(owner, attr, val)
    print(f"  {label:<58s} vs truth64 {rel_l2(y.double(), ref64):.3e}   vs oracle32 {rel_l2(y, ref32):.3e}", flush=True)
    return y


pays, tail = ops.sparse_dil_pays, ops.compact_tail_pays
run("default forms at this batch")
run("forms of a B = 16 batch (list kernels)", ops__sparse_dil_pays=lambda b, n, r, c: pays(16, n, r, c),
    ops__compact_tail_pays=lambda b, n, r, c: tail(16, n, r, c))
run("time embedding concatenated (no per-shape bias)", pvcnn__FP_TEMB_SPLIT=False)
run("first SA level as the operator chain (no recompute kernel)", sa__fuse_mlp=False)
run("bf16x6 convolutions", pv__conv_impl="bf16x6", pv__sparse_gemm="sparse_s3")
run("bf16x6 attention", ops__ATTENTION_IMPL="bf16x6")
run("no hoisted conditioning maps", ops__HOIST_CONDITIONING=False)
run("GroupNorm folding off in the MLPs", mlp__fold_gn=False)
