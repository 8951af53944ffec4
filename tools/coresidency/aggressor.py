"""Background load for the two-process experiments: loops ONE kernel family of the library (or a whole denoiser forward) for a
number of seconds, so that tools/coresidency/two_proc_aggressors.sh can find out WHICH concurrent kernel of a second process disturbs a
victim process (tools/bin/two_proc_repro).   usage: aggressor.py <family> <seconds>"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bdm_amd import ops
from bdm_amd.functional.backend import _backend

family, seconds = sys.argv[1], float(sys.argv[2])
g = torch.Generator().manual_seed(0)
B = 2


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).cuda()


def conv_h2(cin, cout, r):
    x, w, bias = rnd(B, cin, r ** 3), rnd(cout, cin, 3, 3, 3, scale=(27 * cin) ** -0.5), rnd(cout)
    gn = torch.nn.GroupNorm(8, cin).cuda()
    ph, xh = ops.conv3d_h2_pack(w), ops.to_h2(x, gn, swish=True)
    return lambda: ops.conv3d_h2(xh, ph, bias, cin, cout, r)


if family == "forward":
    from bdm_amd.pvd import prepare_pvd_model
    net = prepare_pvd_model({"model": "procedural:1", "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cuda").model
    xt, tt = rnd(B, 3, 1024, scale=0.5), torch.full((B,), 500, dtype=torch.int64, device="cuda")
    fn = lambda: net(xt, tt)
elif family == "conv_h2_32":
    fn = conv_h2(64, 64, 32)
elif family == "conv_h2_16":
    fn = conv_h2(128, 128, 16)
elif family == "conv_h2_8":
    fn = conv_h2(256, 256, 8)
elif family == "to_h2":
    x, gn = rnd(B, 64, 32 ** 3), torch.nn.GroupNorm(8, 64).cuda()
    fn = lambda: ops.to_h2(x, gn, swish=True)
elif family == "attention":
    qkv = rnd(B, 192, 4096, scale=0.3)
    fn = lambda: ops.attention_core(qkv, 64)
elif family == "fps":
    pts = rnd(B, 3, 1024, scale=0.3)
    fn = lambda: _backend.furthest_point_sampling(pts, 256)
elif family == "ball_query":
    pts = rnd(B, 3, 1024, scale=0.3)
    ctr = pts[:, :, :256].contiguous()
    fn = lambda: _backend.ball_query(ctr, pts, 0.2, 32)
elif family == "sparse":
    pts, f, w = rnd(B, 3, 1024, scale=0.3), rnd(B, 64, 1024), rnd(64, 64, 3, 3, 3, scale=1 / 40)
    ops.clear_plan_cache()
    plan, ws3 = ops.voxel_plan(pts, 16), ops.sparse_conv_pack_s3(w)
    fn = lambda: ops.sparse_first_conv_planned(f, plan, ws3, None, 64)
elif family == "sparse_h2":
    pts, f, w = rnd(B, 3, 1024, scale=0.3), rnd(B, 64, 1024), rnd(64, 64, 3, 3, 3, scale=1 / 40)
    ops.clear_plan_cache()
    plan, wh2 = ops.voxel_plan(pts, 16), ops.sparse_conv_pack_h2(w)
    fn = lambda: ops.sparse_first_conv_planned(f, plan, wh2, None, 64)
elif family == "pw":
    x, w, b = rnd(B, 64, 32768), rnd(64, 64, scale=0.15), rnd(64)
    fn = lambda: ops.pointwise_conv(x, w, b)
elif family == "gn":
    x, gn = rnd(B, 64, 32768), torch.nn.GroupNorm(8, 64).cuda()
    fn = lambda: ops.group_norm_(x.clone(), gn.weight, gn.bias, 8, 1e-5, swish=True)
elif family == "torch_matmul":
    a = rnd(4096, 4096)
    fn = lambda: torch.tanh(a @ a * 1e-3)
else:
    raise SystemExit(f"unknown family {family}")
fn(); torch.cuda.synchronize()
t0, n = time.time(), 0
while time.time() - t0 < seconds:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n += 20
print(f"aggressor {family}: {n} iterations in {time.time() - t0:.1f} s")
