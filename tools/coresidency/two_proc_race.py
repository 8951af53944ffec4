"""Root-causing the two-process difference.

Runs REPS forwards of one denoiser on fixed inputs, check-sums EVERY operator output of the forward bit-exactly (every function
of bdm_amd.ops and the plugin backend) and compares each repetition with the first one call by call.  For the first call whose
output differs it prints: the operator, which elements differ (shape, index ranges, run lengths), the good and the bad values,
and where else in the first repetition's outputs the bad bytes occur (a stale or foreign buffer shows up there).

Run two copies at the same time on one GPU (tools/coresidency/two_proc_matrix.sh) under different switches:
   BDM_SIDE_STREAM=0   sampler chain inline (no second stream)      RACE_POINT_STREAM=0  PVConv point branch inline
   RACE_SIDE_PLAN=0    voxel plans on the main stream               RACE_SYNC=1          device-wide sync after every operator
   RACE_MODEL=pc2|pvd  which denoiser                               RACE_B / RACE_N      batch and points
"""
import collections
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from bdm_amd import ops  # noqa: E402
from bdm_amd.functional.backend import _backend, _Backend  # noqa: E402

SYNC = os.environ.get("RACE_SYNC") == "1"
if os.environ.get("RACE_POINT_STREAM") == "0":
    from bdm_amd.modules import PVConv as _PV
    _PV.point_stream = False
if os.environ.get("RACE_SIDE_PLAN") == "0":
    from bdm_amd import pvcnn as _pvcnn
    _pvcnn.SIDE_PLAN = False
HOSTSYNC = os.environ.get("RACE_HOSTSYNC") == "1"  # read every checksum back right away (drains the main stream after every operator)


def finish_log():
    """device scalars -> ints (one transfer per forward)"""
    flat = [v for _, sums in LOG for v in sums if torch.is_tensor(v)]
    host = iter(torch.stack(flat).cpu().tolist()) if flat else iter(())
    out = [(name, tuple(next(host) if torch.is_tensor(v) else v for v in sums)) for name, sums in LOG]
    LOG.clear()
    return out

LOG, OUTS = [], []
KEEP_OUTS = [False]


def leaves(o):
    if torch.is_tensor(o):
        return [o] if (o.is_cuda and o.numel() > 0) else []
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in leaves(x)]
    if isinstance(o, ops.VoxelPlan):
        return [getattr(o, k) for k in ("norm_coords", "vox_coords", "ind", "cnt", "occ_index", "n_occ", "rowocc")]
    return []


def as_int(t):
    t = t.contiguous()
    return t.view({1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()])


def wrap(owner, name, fn):
    def w(*a, **k):
        out = fn(*a, **k)
        if SYNC:
            torch.cuda.synchronize()
        ts = leaves(out[0] if name == "conv3d_h2_gn" else out)  # (its statistics workspace is only partly written)
        sums = [as_int(t).to(torch.int64).sum() for t in ts]   # device scalars: no host sync inside the forward
        if HOSTSYNC:
            sums = [int(v) for v in sums]
        LOG.append((name, sums))
        if KEEP_OUTS[0]:
            OUTS.append([t.detach().cpu() for t in ts])
        return out
    setattr(owner, name, w)


SKIP = ("workspace", "is_point_invariant", "saturation_slot", "poll_h2_saturation", "clear_plan_cache", "h2_activation_scale",
        "amax_slots", "saturation_epoch", "materialize", "attention_h2_ok", "gather_gn_ok")
for name, fn in list(vars(ops).items()):
    if isinstance(fn, types.FunctionType) and not name.startswith("_") and fn.__module__ == ops.__name__ and name not in SKIP:
        wrap(ops, name, fn)
for name in ("furthest_point_sampling", "gather_features_forward", "ball_query", "grouping_forward"):
    wrap(_backend, name, getattr(_Backend, name))

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, N = int(os.environ.get("RACE_B", "2")), int(os.environ.get("RACE_N", "1024"))
which = os.environ.get("RACE_MODEL", "pvd")
g = torch.Generator().manual_seed(3)
tt = torch.full((B,), 500, dtype=torch.int64, device="cuda")
if which == "pvd":
    from bdm_amd.pvd import prepare_pvd_model
    net = prepare_pvd_model({"model": "procedural:1", "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cuda").model
    xt = (torch.randn(B, 3, N, generator=g) * 0.5).cuda()
else:
    from bdm_amd.pvcnn import PVCNN2_PC2
    from bdm_amd.utils.procedural import fill_module_
    net = fill_module_(PVCNN2_PC2(3, 64, extra_feature_channels=387).eval(), seed=5).cuda()
    xt = torch.randn(B, 390, N, generator=g)
    xt[:, :3] *= 0.5
    xt = xt.cuda()

net(xt, tt)  # warm-up: weight packs are cached after the first forward
torch.cuda.synchronize()
KEEP_OUTS[0] = True
LOG.clear()
net(xt, tt)
torch.cuda.synchronize()
KEEP_OUTS[0] = False
log0, outs0 = finish_log(), list(OUTS)
first = {}
shown = 0
for i in range(1, reps):
    LOG.clear()
    net(xt, tt)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(log0, finish_log())):
        if a != b:
            first[i] = (k, a[0])
            break
print(f"model={which} B={B} N={N} env={ {k: v for k, v in os.environ.items() if k.startswith(('BDM_', 'RACE_'))} }: {len(log0)} traced calls per forward; "
      f"repetitions whose trace differs from the first: {len(first)} / {reps - 1}", flush=True)
print("first differing call:", collections.Counter(first.values()).most_common(8), flush=True)
if not first:
    sys.exit(0)

# second phase: repeat until the SAME kind of mismatch happens with outputs kept, then dissect it
target = collections.Counter(v for v in first.values()).most_common(1)[0][0]
for attempt in range(4 * reps):
    LOG.clear(); OUTS.clear()
    KEEP_OUTS[0] = True
    net(xt, tt)
    torch.cuda.synchronize()
    KEEP_OUTS[0] = False
    bad = next((k for k, (a, b) in enumerate(zip(log0, finish_log())) if a != b), None)
    if bad is None:
        continue
    name = log0[bad][0]
    for good_t, bad_t in zip(outs0[bad], OUTS[bad]):
        d = (as_int(good_t) != as_int(bad_t))
        if not bool(d.any()):
            continue
        flat = d.reshape(-1).nonzero().reshape(-1).numpy()
        runs = np.split(flat, np.where(np.diff(flat) != 1)[0] + 1)
        print(f"[attempt {attempt}] call {bad} {name}: tensor {tuple(good_t.shape)} {good_t.dtype}: {len(flat)} elements differ in {len(runs)} run(s); "
              f"runs (start, len, start % 16, byte offset % 128): {[(int(r[0]), len(r), int(r[0]) % 16, int(r[0]) * good_t.element_size() % 128) for r in runs[:6]]}")
        r0 = runs[0]
        gv, bv = good_t.reshape(-1)[r0[0]:r0[0] + len(r0)], bad_t.reshape(-1)[r0[0]:r0[0] + len(r0)]
        print("   good:", [f"{float(v):.6g}" for v in gv[:16]])
        print("   bad :", [f"{float(v):.6g}" for v in bv[:16]])
        idx = np.unravel_index(int(r0[0]), tuple(good_t.shape))
        print("   first bad element index:", tuple(int(v) for v in idx))
        # does the bad run occur elsewhere (bitwise) in the outputs of the reference repetition or of this one?
        pat = as_int(bv).numpy().tobytes()
        if len(pat) >= 16:
            for label, store in (("reference repetition", outs0), ("this repetition", OUTS)):
                for kk, ts in enumerate(store):
                    for t in ts:
                        raw = as_int(t).numpy().tobytes()
                        pos = raw.find(pat)
                        if pos >= 0 and not (store is OUTS and kk == bad):
                            print(f"   the bad bytes occur in {label}, call {kk} {log0[kk][0] if kk < len(log0) else '?'} tensor {tuple(t.shape)} at byte {pos}")
    shown += 1
    if shown >= 3:
        break
