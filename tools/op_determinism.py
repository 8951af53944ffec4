"""Per-operator repeatability under GPU sharing: each operator is run REPS times on fixed inputs and every result is
compared bitwise with the first.  Run two copies at once (tools/op_determinism.sh) to reproduce the multi-process case."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import ops
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(0)
B = 2


def check(name, fn):
    ref = fn()
    ref = [t.clone() for t in (ref if isinstance(ref, (list, tuple)) else [ref])]
    bad = 0
    for _ in range(REPS):
        out = fn()
        out = out if isinstance(out, (list, tuple)) else [out]
        bad += any(not torch.equal(a, b) for a, b in zip(out, ref))
    torch.cuda.synchronize()
    print(f"{name:34s} {bad:4d} / {REPS} repetitions differ", flush=True)


for cin, cout, r in [(64, 64, 32), (128, 128, 16), (256, 256, 8), (32, 32, 32)]:
    x = torch.randn(B, cin, r ** 3, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    gn = torch.nn.GroupNorm(8, cin).cuda()
    ph, ps, pf = ops.conv3d_h2_pack(w), ops.conv3d_s3_pack(w), ops.conv3d_pack(w)
    xh, xs = ops.to_h2(x, gn, swish=True), ops.to_s3(x, gn, swish=True)
    check(f"to_h2 {cin}@{r}", lambda: ops.to_h2(x, gn, swish=True)[0])
    check(f"conv3d_h2 {cin}->{cout}@{r}", lambda: ops.conv3d_h2(xh, ph, bias, cin, cout, r))
    check(f"conv3d_h2_gn {cin}->{cout}@{r}", lambda: list(ops.conv3d_h2_gn(xh, ph, bias, cin, cout, r))[:1])
    check(f"conv3d_s3 {cin}->{cout}@{r}", lambda: ops.conv3d_s3(xs, ps, bias, cin, cout, r))
    check(f"conv3d fp32 {cin}->{cout}@{r}", lambda: ops.conv3d(x, pf, bias, r))
    check(f"group_norm {cin}@{r}", lambda: ops.group_norm_(x.clone(), gn.weight, gn.bias, 8, 1e-5, swish=True))
n = 1024
pts = (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
f = torch.randn(B, 64, n, generator=g).cuda()
w = (torch.randn(64, 64, 3, 3, 3, generator=g) / 40).cuda()
ops.clear_plan_cache()
plan = ops.voxel_plan(pts, 16)
ws3 = ops.sparse_conv_pack_s3(w)
check("sparse first conv (s3) 64->64@16", lambda: ops.sparse_first_conv_planned(f, plan, ws3, None, 64))
qkv = torch.randn(B, 192, 4096, generator=g).cuda() * 0.3
check("attention bf16x6 4096 tokens", lambda: ops.attention_core(qkv, 64))
for c, n, r in [(32, 1024, 32), (64, 1024, 32), (128, 256, 16), (256, 64, 8), (256, 16, 8)]:
    pts = (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
    nc, vc = ops.voxel_coords(pts, r)
    grid = torch.randn(B, c, r ** 3, generator=g).cuda()
    gate = torch.rand(B, c, generator=g).cuda()
    add = torch.randn(B, c, n, generator=g).cuda()
    coef = torch.rand(B, c, 2, generator=g).cuda()
    check(f"devoxelize_gate_add c={c} n={n} r={r}", lambda: ops.devoxelize_gate_add(nc, grid, r, gate=gate, add=add))
    check(f"devoxelize_gn_gate_add c={c} n={n} r={r}", lambda: ops.devoxelize_gn_gate_add(nc, grid, coef, r, gate=gate, add=add))
