"""The voxel branch of a PVConv on voxel LISTS only (csrc/pvconv_compact.hip, round 4) against the dense-grid path it replaces: the
second convolution on the twice-dilated list + 27 class constants == the dense fp16x3 convolution; its GroupNorm statistics, the SE
gate and the devoxelised point features agree at fp32 grade; everything is bit-reproducible."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ops(hip):
    from bdm_amd import ops as o
    return o


@pytest.fixture(params=["0", "256", "64"], autouse=True)
def tile_form(request, monkeypatch, ops):
    """Every test of this file runs on full tiles (round 4's kernel) and on half tiles of 256 / 64 entries (round 6: two workgroups per CU)."""
    from bdm_amd import _lib as L
    if request.param != "0" and not L.has_experimental():
        pytest.skip("half tiles: kernel family of the EXPERIMENTAL=1 build")
    monkeypatch.setattr(ops, "DIL_TILE", request.param)
    ops.clear_plan_cache()
    return request.param


def _layer(ops, cin, cout, r, n, B, seed, spread=0.3):
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(B, cin, n, generator=g).cuda()
    pts = (torch.randn(B, 3, n, generator=g) * spread).cuda()
    pts[-1] *= 0.02                                   # one shape squeezed into a few cells: almost the whole grid is class constants
    w1 = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    w2 = (torch.randn(cout, cout, 3, 3, 3, generator=g) / (27 * cout) ** 0.5).cuda()
    b1, b2 = torch.randn(cout, generator=g).cuda(), torch.randn(cout, generator=g).cuda()
    gn1, gn2 = nn.GroupNorm(8, cout).cuda(), nn.GroupNorm(8, cout).cuda()
    with torch.no_grad():
        for gn in (gn1, gn2):
            gn.weight.copy_(torch.randn(cout, generator=g) * 0.3 + 1.0)
            gn.bias.copy_(torch.randn(cout, generator=g) * 0.2)
    ops.clear_plan_cache()
    plan = ops.voxel_plan(pts, r)
    return f, pts, plan, w1, b1, w2, b2, gn1, gn2


@pytest.mark.parametrize("cin,cout,r,n,B", [(32, 32, 32, 4096, 3), (64, 64, 32, 3000, 2), (128, 64, 16, 1024, 3), (128, 128, 16, 1024, 2),
                                            (192, 128, 8, 256, 3)])
def test_second_convolution_on_the_twice_dilated_list_equals_the_dense_one(ops, cin, cout, r, n, B):
    f, pts, plan, w1, b1, w2, b2, gn1, gn2 = _layer(ops, cin, cout, r, n, B, seed=cin + r + n)
    pk1, pk2 = ops.conv3d_h2_pack(w1), ops.conv3d_h2_pack(w2)
    comp, st1 = ops.sparse_first_conv_os(f, plan, pk1, b1, cout, gn_groups=8, compact=True)
    # dense path: operand split to the H2 grid, dense fp16x3 convolution with statistics
    xh = ops.to_h2(comp, gn1, swish=True, stats=st1)
    y_dense, st_dense = ops.conv3d_h2_gn(xh, pk2, b2, cout, cout, r, 8)
    # list path
    rows_h2, const_h2, const_f32, inv_s = ops.to_h2_rows(comp, plan, gn1, st1)
    assert inv_s == xh[1]
    wsum = ops.conv_class_pack(w2)
    rows, cvals, st_rows = ops.second_conv_rows(rows_h2, const_h2, const_f32, inv_s, plan, pk2, wsum, b2, cout, cout, 8)
    y_list = ops.densify_rows(rows, cvals, plan)
    for b in range(B):
        assert rel(y_list[b], y_dense[b]) < 2e-6, b
    # on the listed voxels both paths run the same MFMA sequence on the same operands: bit-equal there
    idx = plan.d2_index.long()
    for b in range(B):
        v = torch.nonzero(idx[b] >= 0).squeeze(1)
        assert torch.equal(y_list[b][:, v], y_dense[b][:, v]), b
    # statistics of the whole grid (list + class constants) == statistics of the dense output
    ws_d, s_d = st_dense
    pd = ws_d.view(-1).view(torch.float64)[: B * 8 * s_d * 2].view(B, 8, s_d, 2).sum(2)
    pl = st_rows[0].sum(2)
    o = y_dense.double().view(B, 8, -1)
    assert float((pl[..., 0] - o.sum(-1)).abs().max() / o.abs().sum(-1).max()) < 2e-6
    assert float((pl[..., 1] - (o * o).sum(-1)).abs().max() / (o * o).sum(-1).max()) < 2e-6
    assert float((pl - pd).abs().max() / pd.abs().max()) < 2e-6
    # the dense first-convolution output through the list (the hoisted SA0.0 layer keeps a dense grid) gives the same rows
    dense1 = comp.dense()
    rows_d, const_d, cf_d, _ = ops.to_h2_rows(dense1, plan, gn1, st1, bias=b1)
    nd = [int(plan.tile_start[b, :, 1].max()) for b in range(B)]
    for b in range(B):
        assert torch.equal(rows_d[b, :, :, : nd[b]].view(torch.int16), rows_h2[b, :, :, : nd[b]].view(torch.int16))
    assert torch.equal(const_d.view(torch.int16), const_h2.view(torch.int16)) and torch.equal(cf_d, const_f32)
    # bit-reproducible
    rows2, cvals2, st2 = ops.second_conv_rows(rows_h2, const_h2, const_f32, inv_s, plan, pk2, wsum, b2, cout, cout, 8)
    nd2 = [int(plan.d2_tiles[b, :, 1].max()) for b in range(B)]
    assert all(torch.equal(rows2[b, : nd2[b]], rows[b, : nd2[b]]) for b in range(B)) and torch.equal(cvals2, cvals) and torch.equal(st2[0], st_rows[0])


@pytest.mark.parametrize("cin,cout,r,n,B,with_pf", [(32, 32, 32, 4096, 3, True), (64, 64, 32, 3000, 2, False), (128, 128, 16, 1024, 2, True)])
def test_se_gate_and_devoxelisation_from_rows(ops, cin, cout, r, n, B, with_pf):
    f, pts, plan, w1, b1, w2, b2, gn1, gn2 = _layer(ops, cin, cout, r, n, B, seed=7 + cin + r)
    g = torch.Generator().manual_seed(cout)
    pk1, pk2 = ops.conv3d_h2_pack(w1), ops.conv3d_h2_pack(w2)
    comp, st1 = ops.sparse_first_conv_os(f, plan, pk1, b1, cout, gn_groups=8, compact=True)
    xh = ops.to_h2(comp, gn1, swish=True, stats=st1)
    y_dense, st_dense = ops.conv3d_h2_gn(xh, pk2, b2, cout, cout, r, 8)
    rows_h2, const_h2, const_f32, inv_s = ops.to_h2_rows(comp, plan, gn1, st1)
    rows, cvals, st_rows = ops.second_conv_rows(rows_h2, const_h2, const_f32, inv_s, plan, pk2, ops.conv_class_pack(w2), b2, cout, cout, 8)
    hid = max(cout // 8, 4)
    sw1 = (torch.randn(hid, cout, generator=g) / cout ** 0.5).cuda()
    sw2 = (torch.randn(cout, hid, generator=g) / hid ** 0.5).cuda()
    add = torch.randn(B, cout, n, generator=g).cuda()
    pf = None
    if with_pf:   # raw point-branch output + the statistics its 1x1 convolution would leave
        pgn = nn.GroupNorm(8, cout).cuda()
        with torch.no_grad():
            pgn.weight.copy_(torch.randn(cout, generator=g) * 0.3 + 1.0)
        a = add.double().view(B, 8, -1)
        part = torch.stack([a.sum(-1), (a * a).sum(-1)], -1).view(B, 8, 1, 2).contiguous()
        pf = ((part, 1, 8), pgn)
    if pf is not None:
        gate_d, coef_d, pfc_d = ops.se_gate_gn(y_dense, st_dense, gn2, sw1, sw2, pf=pf, n_points=n)
        gate_r, coef_r, pfc_r = ops.se_gate_gn_rows(rows, cvals, plan, st_rows, gn2, sw1, sw2, pf=pf, n_points=n)
        assert torch.equal(pfc_d, pfc_r)
    else:
        gate_d, coef_d = ops.se_gate_gn(y_dense, st_dense, gn2, sw1, sw2)
        gate_r, coef_r = ops.se_gate_gn_rows(rows, cvals, plan, st_rows, gn2, sw1, sw2)
        pfc_d = pfc_r = None
    assert rel(coef_r, coef_d) < 2e-6 and rel(gate_r, gate_d) < 2e-6
    out_d = ops.devoxelize_gn_gate_add(plan.norm_coords, y_dense, coef_d, r, gate=gate_d, add=add, add_coef=pfc_d)
    out_r = ops.devoxelize_gn_gate_add_rows(plan.norm_coords, rows, cvals, plan, coef_r, gate=gate_r, add=add, add_coef=pfc_r)
    assert bool(torch.isfinite(out_r).all()) and rel(out_r, out_d) < 5e-6
    # with the dense path's coefficients and gate the gather itself is the same arithmetic on the same values: bit-equal
    assert torch.equal(ops.devoxelize_gn_gate_add_rows(plan.norm_coords, rows, cvals, plan, coef_d, gate=gate_d, add=add, add_coef=pfc_d),
                       ops.devoxelize_gn_gate_add(plan.norm_coords, y_dense, coef_d, r, gate=gate_d, add=add, add_coef=pfc_d))
    assert torch.equal(out_r, ops.devoxelize_gn_gate_add_rows(plan.norm_coords, rows, cvals, plan, coef_r, gate=gate_r, add=add, add_coef=pfc_r))


@pytest.mark.parametrize("cin,cout,r,n", [(32, 32, 32, 4096), (64, 64, 32, 2500), (128, 64, 16, 1024), (256, 256, 8, 64), (390, 32, 32, 4096)])
def test_pvconv_on_voxel_lists_equals_pvconv_on_grids(ops, monkeypatch, cin, cout, r, n):
    """The whole module: list path (forced) vs dense-grid path (forced off); with the first convolution in either form."""
    from bdm_amd.modules import PVConv
    from bdm_amd.utils.procedural import fill_module_
    pv = fill_module_(PVConv(cin, cout, 3, resolution=r, with_se=True, with_se_relu=True).eval(), seed=cin + r).cuda()
    g = torch.Generator().manual_seed(n)
    B = 3
    f, c = torch.randn(B, cin, n, generator=g).cuda(), (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
    t = torch.zeros(B, 8, n, device="cuda")
    monkeypatch.setattr(PVConv, "compact_tail_resolutions", {8, 16, 32})
    monkeypatch.setattr(PVConv, "sparse_dil_resolutions", {8, 16, 32})
    outs = {}
    for first in (True, False):
        monkeypatch.setattr(PVConv, "sparse_dil_always", first)
        monkeypatch.setattr(PVConv, "sparse_conv", "dil" if first else "gemm")
        for tail in ("always", "0"):
            monkeypatch.setattr(PVConv, "compact_tail", tail)
            ops.clear_plan_cache()
            outs[(first, tail)] = pv((f, c, t))[0].clone()
            ops.clear_plan_cache()
            assert torch.equal(pv((f, c, t))[0], outs[(first, tail)])          # deterministic
    ref = outs[(False, "0")]                                                   # GEMM + gather, dense grids: the round-3 path
    for k, o in outs.items():
        assert bool(torch.isfinite(o).all()) and rel(o, ref) < 1e-5, (k, rel(o, ref))
