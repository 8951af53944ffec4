"""The fused glue of a PVConv on the small voxel grids (csrc/pvconv_small.hip, round 5) against the operator chain it replaces:

* bdm_pvconv_tail_small (SE gate + GroupNorm-2 + Swish + devoxelisation + point branch) is BIT-identical to bdm_se_gate_gn_pf +
  bdm_devoxelize_gn_gate_add_pf;
* its head (the next PVConv's first-convolution operand) holds the values of bdm_sparse_voxel_features_f32 at the split's precision;
* bdm_sparse_conv_gather_h2_small equals gather + to_h2_stats at fp32 grade;
* a chain of PVConvs (the FP0 / FP1 stages: 256 channels at 8^3, 64 / 256 points) gives the same features with the glue on and off,
  both against the CPU oracle, is bit-reproducible, and a shape's bits do not depend on its batch.
"""
import pytest
import torch
import torch.nn as nn

from helpers import experimental, parity, current_test

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ops(hip):
    from bdm_amd import ops as o
    return o


def _gn(c, g, seed):
    gn = nn.GroupNorm(8, c).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.randn(c, generator=g) * 0.3 + 1.0)
        gn.bias.copy_(torch.randn(c, generator=g) * 0.2)
    return gn


def _h2_values(xh, inv_scale):
    """(B, G, 2, V, 8) fp16 records -> (B, G * 8, V) fp32 values hi + lo, unscaled"""
    v = (xh[:, :, 0].float() + xh[:, :, 1].float()) * inv_scale
    B, G, V, _ = v.shape
    return v.permute(0, 1, 3, 2).reshape(B, G * 8, V)


@pytest.mark.parametrize("c,r,n,B,with_pf", [(256, 8, 64, 3, True), (256, 8, 256, 2, True), (128, 8, 256, 3, False), (64, 8, 100, 2, True),
                                             (256, 8, 130, 2, True), (256, 8, 300, 2, True), (512, 8, 64, 2, False)])   # the last two: generic kernel
def test_fused_tail_is_bit_identical_to_gate_plus_devoxelisation(ops, c, r, n, B, with_pf):
    g = torch.Generator().manual_seed(c + n)
    grid = torch.randn(B, c, r ** 3, generator=g).cuda()
    pts = (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
    ops.clear_plan_cache()
    plan = ops.voxel_plan(pts, r)
    gn2 = _gn(c, g, 0)
    a = grid.double().view(B, 8, -1)
    ws = torch.zeros(ops.L.lib().bdm_group_norm_workspace_bytes(B, 8), dtype=torch.uint8, device="cuda")
    part = torch.stack([a.sum(-1), (a * a).sum(-1)], -1).view(B, 8, 1, 2).contiguous()
    ws.view(torch.float64)[: part.numel()] = part.view(-1)
    stats = (ws, 1)
    hid = c // 8
    w1 = (torch.randn(hid, c, generator=g) / c ** 0.5).cuda()
    w2 = (torch.randn(c, hid, generator=g) / hid ** 0.5).cuda()
    add = torch.randn(B, c, n, generator=g).cuda()
    pf = None
    if with_pf:
        pgn = _gn(c, g, 1)
        pa = add.double().view(B, 8, -1)
        pf = ((torch.stack([pa.sum(-1), (pa * pa).sum(-1)], -1).view(B, 8, 1, 2).contiguous(), 1, 8), pgn)
        gate, coef, pfc = ops.se_gate_gn(grid, stats, gn2, w1, w2, pf=pf, n_points=n)
        mean, coef2, pfc2 = ops.se_means_gn(grid, stats, gn2, pf=pf, n_points=n)
        assert torch.equal(pfc, pfc2)
    else:
        gate, coef = ops.se_gate_gn(grid, stats, gn2, w1, w2)
        mean, coef2 = ops.se_means_gn(grid, stats, gn2)
        pfc = None
    assert torch.equal(coef, coef2)
    ref = ops.devoxelize_gn_gate_add(plan.norm_coords, grid, coef, r, gate=gate, add=add, add_coef=pfc)
    out, rows = ops.pvconv_tail_small(plan.norm_coords, grid, coef, mean, w1, w2, r, add=add, add_coef=pfc)
    assert rows is None and torch.equal(out, ref)
    # ... and with the head: same features, plus the next PVConv's operand = the per-cell means of those features
    sat = torch.zeros(1, dtype=torch.int32, device="cuda")
    scale = 2.0 ** 7
    out2, rows = ops.pvconv_tail_small(plan.norm_coords, grid, coef, mean, w1, w2, r, add=add, add_coef=pfc, head=(plan, scale, sat))
    assert torch.equal(out2, ref) and int(sat) == 0
    xr = torch.empty(B, c // 8, plan.n_max, 8, dtype=torch.float32, device="cuda")
    amax = torch.zeros(B, dtype=torch.float32, device="cuda")
    f, _, _, _, bs_f, ld_f = ops._bcl(ref)
    L = ops.L
    L.check(L.lib().bdm_sparse_voxel_features_f32(B, c, n, r, plan.n_max, L.ptr(f), bs_f, ld_f, L.ptr(plan.cnt), L.ptr(plan.ws),
                                                  L.ptr(plan.occ_list), L.ptr(plan.n_occ), L.ptr(xr), L.ptr(amax), L.stream()), "features")
    want = xr.permute(0, 1, 3, 2).reshape(B, c, plan.n_max)          # (B, G, n_max, 8) -> (B, C, n_max)
    got = _h2_values(rows.xh, 1.0 / scale)
    assert float((got - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -20   # two fp16 terms: >= 21 bits of the largest value
    assert rel(got, want) < 5e-7
    # the GEMM derives exactly this scale from amax_out: act_scale_from_max(24576 / s) == s
    assert torch.equal(rows.amax, torch.full((B,), 24576.0 / scale, device="cuda"))
    # a value beyond fp16's range raises the layer's sticky word
    _, _ = ops.pvconv_tail_small(plan.norm_coords, grid, coef, mean, w1, w2, r, add=add, add_coef=pfc, head=(plan, 2.0 ** 20, sat))
    assert int(sat) == 1


@experimental   # csrc/experimental/sparse_gather_h2_small.hip (measured not faster: DESIGN.md 7.9)
@pytest.mark.parametrize("cin,cout,r,n,B", [(256, 256, 8, 64, 3), (256, 256, 8, 256, 2), (192, 128, 8, 256, 3)])
def test_gather_with_groupnorm_and_operand_split_equals_gather_then_to_h2(ops, cin, cout, r, n, B):
    g = torch.Generator().manual_seed(cin + n)
    f = torch.randn(B, cin, n, generator=g).cuda()
    pts = (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
    pts[-1] *= 0.05                                    # one shape squeezed into a few cells: most of its grid is pure bias
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    gn1 = _gn(cout, g, 0)
    ops.clear_plan_cache()
    plan = ops.voxel_plan(pts, r)
    pk = ops.sparse_conv_pack_h2(w)
    v, st = ops.sparse_first_conv_planned(f, plan, pk, bias, cout, gn_groups=8)
    sat = torch.zeros(1, dtype=torch.int32, device="cuda")
    xh_ref, inv_ref = ops.to_h2(v, gn1, swish=True, saturated=sat, stats=st)
    xh, inv = ops.sparse_first_conv_planned(f, plan, pk, bias, cout, h2_out=(gn1, ops.h2_activation_scale(gn1), sat))
    assert inv == inv_ref and int(sat) == 0
    a, b = _h2_values(xh, inv), _h2_values(xh_ref, inv_ref)
    parity(current_test() + " operand of the second convolution", rel(a, b), 2e-6)
    assert bool(torch.isfinite(a).all()) and rel(a, b) < 2e-6
    # against fp64: Swish(GroupNorm(conv1 output)) from the dense fp32 grid
    y = v.double().view(B, 8, -1)
    mu, var = y.mean(-1, keepdim=True), y.var(-1, unbiased=False, keepdim=True)
    z = ((y - mu) / (var + gn1.eps).sqrt()).view(B, cout, -1) * gn1.weight.double()[None, :, None] + gn1.bias.double()[None, :, None]
    ref = (z * torch.sigmoid(z)).float()
    assert rel(a, ref) < 2e-6
    # bit-reproducible
    xh2, _ = ops.sparse_first_conv_planned(f, plan, pk, bias, cout, h2_out=(gn1, ops.h2_activation_scale(gn1), sat))
    assert torch.equal(xh2.view(torch.int16), xh.view(torch.int16))


def _stage(cin, c, r, blocks, seed):
    from bdm_amd.modules import PVConv
    from bdm_amd.utils.procedural import fill_module_
    mods = [PVConv(cin if i == 0 else c, c, 3, resolution=r, with_se=True, with_se_relu=True) for i in range(blocks)]
    return fill_module_(nn.Sequential(*mods).eval(), seed=seed).cuda()


@pytest.mark.parametrize("tail_only", [True, pytest.param(False, marks=experimental)])
@pytest.mark.parametrize("cin,c,n,blocks", [(256, 256, 64, 3), (256, 256, 256, 3), (192, 128, 256, 1), (256, 256, 300, 2)])
def test_pvconv_chain_with_and_without_the_fused_glue(ops, oracle_ops, monkeypatch, cin, c, n, blocks, tail_only):
    """FP0 (64 points), FP1 (256 points) and SA2.0 stages at 8^3: glue on == glue off at fp32 grade, both == the CPU oracle; the handed-on
    operand is really used (the feature pass of PVConvs 2.. is skipped); deterministic; batch-invariant."""
    from bdm_amd import pvcnn
    from oracle import ref_net
    r, B = 8, 3
    seq = _stage(cin, c, r, blocks, seed=cin + n)
    g = torch.Generator().manual_seed(n + blocks)
    f = torch.randn(B, cin, n, generator=g).cuda()
    co = (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
    t = torch.zeros(B, 8, n, device="cuda")

    def run(ff, cc):
        ops.clear_plan_cache()
        return pvcnn.run_blocks(seq, (ff, cc, t[: ff.shape[0]]))[0].clone()

    monkeypatch.setattr(ops, "SMALL_GLUE", True)
    monkeypatch.setattr(ops, "SMALL_GLUE_TAIL_ONLY", tail_only)
    on = run(f, co)
    assert torch.equal(run(f, co), on)                                   # deterministic
    for b in range(B):                                                   # a shape's bits do not depend on its batch
        assert torch.equal(run(f[b:b + 1].contiguous(), co[b:b + 1].contiguous())[0], on[b]), b
    monkeypatch.setattr(ops, "SMALL_GLUE", False)
    off = run(f, co)
    parity(current_test() + " glue on vs off", rel(on, off), 5e-6)
    assert bool(torch.isfinite(on).all()) and rel(on, off) < 5e-6
    # the oracle: the reference's PVConv, block by block
    sd = {k: v.detach().cpu() for k, v in seq.state_dict().items()}
    x = f.cpu()
    for i in range(blocks):
        x = ref_net.pvconv(sd, f"{i}.", x, co.cpu(), r, False)
    parity(current_test() + " glue on vs oracle", rel(on.cpu(), x), 1e-5)
    parity(current_test() + " glue off vs oracle", rel(off.cpu(), x), 1e-5)
    assert rel(on.cpu(), x) < 1e-5 and rel(off.cpu(), x) < 1e-5


def test_the_next_pvconv_really_takes_the_handed_on_operand(ops, monkeypatch):
    from bdm_amd import pvcnn
    seq = _stage(256, 256, 8, 3, seed=5)
    g = torch.Generator().manual_seed(1)
    f, co = torch.randn(2, 256, 64, generator=g).cuda(), (torch.randn(2, 3, 64, generator=g) * 0.3).cuda()
    t = torch.zeros(2, 8, 64, device="cuda")
    seen = []
    real = ops.sparse_first_conv_planned
    monkeypatch.setattr(ops, "sparse_first_conv_planned", lambda *a, **k: (seen.append((k.get("rows") is not None, k.get("h2_out") is not None)), real(*a, **k))[1])
    monkeypatch.setattr(ops, "SMALL_GLUE", True)
    monkeypatch.setattr(ops, "SMALL_GLUE_TAIL_ONLY", True)
    ops.clear_plan_cache()
    pvcnn.run_blocks(seq, (f, co, t))
    assert seen == [(False, False), (True, False), (True, False)]   # (operand handed on, fused gather): the default mode
    # the last PVConv of a Sequential has no successor: nothing is handed on
    assert seq[2]._next_pv is None and seq[0]._next_pv is seq[1]
    assert "_next_pv" not in dict(seq[0].named_modules()) and len(list(seq[0].state_dict())) == len(list(seq[2].state_dict()))
