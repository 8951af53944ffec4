"""Module-level fixtures produced by the REFERENCE's own modules and functional wrappers (tests/golden/modules.npz, written
by oracle/gen_golden_modules.py with /root/reference imported in the build container): one SharedMLP (1-D and 2-D), SE3d,
Attention (3-D and 1-D), Voxelization, PVConv (with and without attention), BallQuery, PointNetSAModule, PointNetFPModule
and each of the six functional wrappers.
  * CPU tests: the oracle's restatement (oracle/ref_net.py, oracle/ops.py) reproduces them;
  * GPU tests (-m gpu): the HIP modules of bdm_amd/modules.py and bdm_amd/functional reproduce them.
Indices are compared bit-exactly, floats to 1e-5 relative L2 (HIP: fp32-grade split products)."""
import pytest
import torch

import module_fixtures as MF
from helpers import rel_l2

TOL = 1e-5


@pytest.fixture(scope="module")
def gold():
    return MF.load(), MF.make_inputs()


def _centers(g):
    return MF.outs(g, "f_furthest_point_sample")[0]


def _vox_in(x):
    return torch.clamp((x["coords"] * 2.0 + 0.5) * MF.R, 0, MF.R - 1)


# ---------------------------------------------------------------------------------------------------------------
# CPU: oracle vs the reference's modules
# ---------------------------------------------------------------------------------------------------------------
def test_oracle_functional_wrappers(gold, oracle_ops):
    g, x = gold
    O = oracle_ops
    idx = O.furthest_point_sampling(x["coords"], MF.M)
    centers = O.gather_features_forward(x["coords"], idx)
    assert torch.equal(centers, _centers(g))
    nbr = O.ball_query(centers, x["coords"], MF.RADIUS, MF.U)
    assert torch.equal(nbr, MF.outs(g, "f_ball_query")[0])
    assert torch.equal(O.grouping_forward(x["feat32"], nbr), MF.outs(g, "f_grouping")[0])
    assert torch.equal(O.three_nearest_neighbors_interpolate_forward(x["coords"], centers, x["cfeat"])[0],
                       MF.outs(g, "f_nearest_neighbor_interpolate")[0])
    vi = _vox_in(x)
    vox = O.avg_voxelize_forward(x["feat16"], torch.round(vi).to(torch.int32).contiguous(), MF.R)[0]
    assert torch.equal(vox.view(MF.B, 16, MF.R, MF.R, MF.R), MF.outs(g, "f_avg_voxelize")[0])
    dv = O.trilinear_devoxelize_forward(MF.R, False, vi.contiguous(), x["grid_r"])[0]
    assert torch.equal(dv, MF.outs(g, "f_trilinear_devoxelize")[0])


def test_oracle_modules(gold, oracle_ops):
    from oracle import ref_net as RN
    g, x = gold
    sd = lambda n: {n + "." + k: v for k, v in MF.state_dict(g, n).items()}  # noqa: E731
    assert rel_l2(RN.shared_mlp(sd("shared_mlp_1d"), "shared_mlp_1d.", x["feat16"]), MF.outs(g, "shared_mlp_1d")[0]) < TOL
    assert rel_l2(RN.shared_mlp(sd("shared_mlp_2d"), "shared_mlp_2d.", x["grouped16"]), MF.outs(g, "shared_mlp_2d")[0]) < TOL
    assert rel_l2(RN.attention(sd("attention_3d"), "attention_3d.", x["grid"]), MF.outs(g, "attention_3d")[0]) < TOL
    assert rel_l2(RN.attention(sd("attention_1d"), "attention_1d.", x["feat32"][:, :, :16].contiguous()),
                  MF.outs(g, "attention_1d")[0]) < TOL
    nc, vc = RN.voxel_coords(x["coords"], MF.R)
    vox_ref, nc_ref = MF.outs(g, "voxelization")
    assert torch.equal(nc, nc_ref)
    vox = oracle_ops.avg_voxelize_forward(x["feat16"], vc.contiguous(), MF.R)[0]
    assert torch.equal(vox.view_as(vox_ref), vox_ref)
    for name, att in (("pvconv_plain", False), ("pvconv_attention", True)):
        y = RN.pvconv(sd(name), name + ".", x["feat16"], x["coords"], MF.R, att)
        assert rel_l2(y, MF.outs(g, name)[0]) < TOL, name
    f, c, t = RN.sa_module(sd("sa_module"), "sa_module.", x["feat32"], x["coords"], x["temb"], MF.M, MF.RADIUS, MF.U)
    rf, rc, rt = MF.outs(g, "sa_module")
    assert rel_l2(f, rf) < TOL and torch.equal(c, rc) and torch.equal(t, rt)
    f, t = RN.fp_module(sd("fp_module"), "fp_module.", x["coords"], _centers(g), x["cfeat"], x["feat16"], x["ctemb"])
    rf, rt = MF.outs(g, "fp_module")
    assert rel_l2(f, rf) < TOL and torch.equal(t, rt)


# ---------------------------------------------------------------------------------------------------------------
# GPU: HIP modules vs the reference's modules
# ---------------------------------------------------------------------------------------------------------------
def _hip_module(g, name, module):
    module = module.eval()
    missing = module.load_state_dict(MF.state_dict(g, name), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(module.state_dict().keys()) == [str(k) for k in g[name + "__keys"]]   # same keys, same order
    return module.cuda()


@pytest.mark.gpu
def test_hip_functional_wrappers(gold, hip):
    from bdm_amd import functional as F
    g, x = gold
    d = {k: v.cuda() for k, v in x.items()}
    centers = F.furthest_point_sample(d["coords"], MF.M)
    assert torch.equal(centers.cpu(), _centers(g))
    nbr = F.ball_query(centers, d["coords"], MF.RADIUS, MF.U)
    assert torch.equal(nbr.cpu(), MF.outs(g, "f_ball_query")[0])
    assert torch.equal(F.grouping(d["feat32"], nbr).cpu(), MF.outs(g, "f_grouping")[0])
    assert torch.equal(F.nearest_neighbor_interpolate(d["coords"], centers, d["cfeat"]).cpu(),
                       MF.outs(g, "f_nearest_neighbor_interpolate")[0])
    vi = _vox_in(x).cuda()
    assert torch.equal(F.avg_voxelize(d["feat16"], torch.round(vi).to(torch.int32), MF.R).cpu(), MF.outs(g, "f_avg_voxelize")[0])
    assert torch.equal(F.trilinear_devoxelize(d["grid_r"], vi, MF.R, False).cpu(),
                       MF.outs(g, "f_trilinear_devoxelize")[0])


@pytest.mark.gpu
def test_hip_dense_modules(gold, hip):
    from bdm_amd.modules import SE3d, Attention, SharedMLP
    g, x = gold
    d = {k: v.cuda() for k, v in x.items()}
    m = _hip_module(g, "shared_mlp_1d", SharedMLP(16, [24, 32], dim=1))
    assert rel_l2(m(d["feat16"]).cpu(), MF.outs(g, "shared_mlp_1d")[0]) < TOL
    m = _hip_module(g, "shared_mlp_2d", SharedMLP(16, [24, 32], dim=2))
    assert rel_l2(m(d["grouped16"]).cpu(), MF.outs(g, "shared_mlp_2d")[0]) < TOL
    se = _hip_module(g, "se3d", SE3d(16, use_relu=True))
    gate = se.gate(d["grid"])                        # the product multiplies inside the devoxelisation gather
    assert rel_l2((d["grid"] * gate.view(MF.B, 16, 1, 1, 1)).cpu(), MF.outs(g, "se3d")[0]) < TOL
    a3 = _hip_module(g, "attention_3d", Attention(16, 8, D=3))
    assert rel_l2(a3(d["grid"]).cpu(), MF.outs(g, "attention_3d")[0]) < TOL
    a1 = _hip_module(g, "attention_1d", Attention(32, 8, D=1))
    assert rel_l2(a1(d["feat32"][:, :, :16].contiguous()).cpu(), MF.outs(g, "attention_1d")[0]) < TOL


@pytest.mark.gpu
def test_hip_point_voxel_modules(gold, hip):
    from bdm_amd.modules import BallQuery, PointNetFPModule, PointNetSAModule, PVConv, Voxelization
    g, x = gold
    d = {k: v.cuda() for k, v in x.items()}
    vox, nc = Voxelization(MF.R)(d["feat16"], d["coords"])
    vox_ref, nc_ref = MF.outs(g, "voxelization")
    # normalised coordinates: the per-shape mean / max-norm reductions (voxelization.py:18-20) have no defined summation
    # order (torch's CPU and CUDA reductions differ from each other too), so agreement is to the last bits, not bitwise
    assert float((nc.cpu() - nc_ref).abs().max()) < 4e-6                          # values in [0, R-1] = [0, 7]
    assert rel_l2(vox.cpu().view_as(vox_ref), vox_ref) < 1e-6                     # no rounding flip on this fixture
    for name, att in (("pvconv_plain", False), ("pvconv_attention", True)):
        m = _hip_module(g, name, PVConv(16, 32, 3, resolution=MF.R, attention=att, with_se=True, with_se_relu=True))
        f, c, t = m((d["feat16"], d["coords"], d["temb"]))
        assert rel_l2(f.cpu(), MF.outs(g, name)[0]) < TOL, name
    centers = _centers(g).cuda()
    gr, gt = BallQuery(MF.RADIUS, MF.U)(d["coords"], centers, d["temb"], d["feat32"])
    rg, rgt = MF.outs(g, "ball_query_module")
    assert torch.equal(gr.cpu(), rg) and torch.equal(gt.cpu(), rgt)
    sa = _hip_module(g, "sa_module", PointNetSAModule(MF.M, MF.RADIUS, MF.U, in_channels=32, out_channels=[32, 48]))
    f, c, t = sa((d["feat32"], d["coords"], d["temb"]))
    rf, rc, rt = MF.outs(g, "sa_module")
    assert rel_l2(f.cpu(), rf) < TOL and torch.equal(c.cpu(), rc) and torch.equal(t.cpu(), rt)
    fp = _hip_module(g, "fp_module", PointNetFPModule(in_channels=32 + 16, out_channels=[32, 24]))
    f, c, t = fp((d["coords"], centers, d["cfeat"], d["feat16"], d["ctemb"]))
    rf, rt = MF.outs(g, "fp_module")
    assert rel_l2(f.cpu(), rf) < TOL and rel_l2(t.cpu(), rt) < 1e-6   # interpolated t_emb: sum of weights ~ 1
