"""Which operator makes a shape's result depend on its batch?  (tests/test_hip_full_size.py: a shape run at B = 1 must give the bits
it gives inside a batch.)  Runs a denoiser forward at batch B and again on shape ROW alone, check-sums row ROW of every tensor that
goes into and comes out of every bdm_amd.ops / plugin-backend call, and lists the calls whose INPUTS agree bit for bit while their
OUTPUTS do not -- the batch-dependent operators themselves, not the ones downstream of them.

    python tools/batch_invariance_trace.py [pc2|pvd] [B] [N] [row]
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from bdm_amd import ops  # noqa: E402
from bdm_amd.functional.backend import _backend, _Backend  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "pvd"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ROW = int(sys.argv[4]) if len(sys.argv) > 4 else B - 1
STATE = {"B": B, "row": ROW}
LOG = []


class Stats:
    """GroupNorm slice partials (partial, slices, groups): only the first B * groups * slices * 2 doubles are meaningful"""

    def __init__(self, partial, slices, groups):
        self.partial, self.slices, self.groups = partial, int(slices), int(groups)


def leaves(o):
    if torch.is_tensor(o):
        return [o] if (o.is_cuda and o.numel() > 0) else []
    if isinstance(o, tuple) and len(o) == 3 and torch.is_tensor(o[0]) and isinstance(o[1], int) and isinstance(o[2], int):
        return [Stats(*o)]
    if isinstance(o, tuple) and len(o) == 2 and torch.is_tensor(o[0]) and o[0].dtype == torch.uint8 and isinstance(o[1], int):
        return [Stats(o[0].view(torch.float64), o[1], 8)]   # conv3d_h2_gn's (workspace, slices); GroupNorm(8)
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in leaves(x)]
    if isinstance(o, dict):
        return [t for x in o.values() for t in leaves(x)]
    if isinstance(o, ops.VoxelPlan):
        return [getattr(o, k) for k in ("norm_coords", "vox_coords", "ind", "cnt", "occ_index", "n_occ", "rowocc")]
    return []


def rowsum(t):
    if isinstance(t, Stats):
        per = t.groups * t.slices * 2
        flat = t.partial.reshape(-1).view(torch.float64) if t.partial.dtype != torch.float64 else t.partial.reshape(-1)
        t = flat[STATE["row"] * per:(STATE["row"] + 1) * per]
        return (("stats", per), t.contiguous().view(torch.int64).sum())
    if t.dim() == 1 and t.numel() == STATE["B"] and STATE["B"] > 1:
        t = t[STATE["row"]:STATE["row"] + 1]
    if t.dim() >= 2 and t.shape[0] == STATE["B"]:
        t = t[STATE["row"]]
    elif t.dim() == 1 and STATE["B"] > 1 and t.numel() % STATE["B"] == 0 and t.numel() >= 8 * STATE["B"]:
        t = t.view(STATE["B"], -1)[STATE["row"]]   # flat per-shape workspaces (GroupNorm slice partials, ...)
    t = t.contiguous()
    v = t.view({1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()])
    return (tuple(t.shape), v.to(torch.int64).sum())


def wrap(owner, name, fn):
    def w(*a, **k):
        ins = [rowsum(t) for t in leaves((a, k))]
        out = fn(*a, **k)
        outs = [rowsum(t) for t in leaves(out)]
        LOG.append((name, ins, outs))
        return out
    setattr(owner, name, w)


SKIP = ("workspace", "is_point_invariant", "saturation_slot", "poll_h2_saturation", "clear_plan_cache", "h2_activation_scale",
        "amax_slots", "saturation_epoch", "materialize", "attention_h2_ok", "gather_gn_ok")
for name, fn in list(vars(ops).items()):
    if isinstance(fn, types.FunctionType) and not name.startswith("_") and fn.__module__ == ops.__name__ and name not in SKIP:
        wrap(ops, name, fn)
for name in ("furthest_point_sampling", "gather_features_forward", "ball_query", "grouping_forward"):
    wrap(_backend, name, getattr(_Backend, name))

from helpers import point_cloud_inputs  # noqa: E402
from bdm_amd.pvcnn import PVCNN2_PC2, PVCNN2_PVD  # noqa: E402
from bdm_amd.utils.procedural import fill_module_  # noqa: E402
extra = 387 if which == "pc2" else 0
net = fill_module_((PVCNN2_PC2 if extra else PVCNN2_PVD)(3, 64, extra_feature_channels=extra).eval(), seed=31 + B).cuda()
x = point_cloud_inputs(B, 3 + extra, N, seed=7000 + N + B)
t = (torch.arange(B) * 31 + 5) % 1000


def run(xx, tt, b, row):
    STATE["B"], STATE["row"] = b, row
    LOG.clear()
    y = net(xx.cuda(), tt.cuda())
    torch.cuda.synchronize()
    log = [(n, [(s, int(v)) for s, v in i], [(s, int(v)) for s, v in o]) for n, i, o in LOG]
    return y.cpu(), log


net(x.cuda(), t.cuda())  # warm-up (weight packs)
yb, logb = run(x, t, B, ROW)
y1, log1 = run(x[ROW:ROW + 1].contiguous(), t[ROW:ROW + 1], 1, 0)
print(f"{which}: B={B} N={N} row={ROW}: {len(logb)} / {len(log1)} traced calls; final outputs equal: {torch.equal(yb[ROW:ROW + 1], y1)}; "
      f"rel-L2 {float((yb[ROW:ROW + 1] - y1).norm() / y1.norm()):.3e}")
if [n for n, _, _ in logb] != [n for n, _, _ in log1]:
    print("!! the two runs make DIFFERENT call sequences (batch-dependent Python dispatch):")
    for k, (a, b) in enumerate(zip(logb, log1)):
        if a[0] != b[0]:
            print(f"   first divergence at call {k}: batch run {a[0]}, single run {b[0]}")
            break
culprits = 0
for k, (a, b) in enumerate(zip(logb, log1)):
    if a[0] != b[0]:
        break
    ins_equal = [s for s, _ in a[1]] == [s for s, _ in b[1]] and all(x[1] == y[1] for x, y in zip(a[1], b[1]))
    # shapes of batch-leading tensors differ in dim 0 only after the row slice, so the shape lists compare equal when invariant
    outs_equal = [s for s, _ in a[2]] == [s for s, _ in b[2]] and all(x[1] == y[1] for x, y in zip(a[2], b[2]))
    if ins_equal and not outs_equal:
        culprits += 1
        shapes_b, shapes_1 = [s for s, _ in a[2]], [s for s, _ in b[2]]
        print(f"   call {k:3d} {a[0]}: inputs equal, outputs differ; output shapes batch {shapes_b} single {shapes_1}; "
              f"differing outputs: {[i for i, (x, y) in enumerate(zip(a[2], b[2])) if x != y]}")
print(f"{culprits} batch-dependent operator call(s)")
