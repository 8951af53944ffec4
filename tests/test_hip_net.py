"""GPU parity of the full denoisers on the HIP path:
  (1) against the golden vectors produced by the REFERENCE's own nn.Modules (tests/golden/*.npz);
  (2) against the CPU oracle on fresh seeded inputs.
Tolerance: 1e-4 relative L2 per forward (the north star allows 1e-3 on the final cloud)."""
import os

import numpy as np
import pytest
import torch

from helpers import check_rel_l2, experimental, point_cloud_inputs, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def build(cls, g, **kw):
    from bdm_amd.utils.procedural import fill_module_
    net = cls(num_classes=3, embed_dim=64, **kw).eval()
    fill_module_(net, seed=int(g["weight_seed"]))
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]  # checkpoint compatibility
    return net.cuda()


def test_pc2_reduced_width_golden(hip):
    from bdm_amd.pvcnn import PVCNN2_PC2
    g = load("pc2_wm025_n1100.npz")
    net = build(PVCNN2_PC2, g, extra_feature_channels=int(g["S"]), width_multiplier=0.25)
    x = point_cloud_inputs(int(g["B"]), 3 + int(g["S"]), int(g["N"]), int(g["input_seed"]))
    y = net(x.cuda(), torch.from_numpy(g["t"]).cuda()).cpu()
    check_rel_l2(y, torch.from_numpy(g["out"]), TOL)


def test_pc2_full_golden(hip):
    from bdm_amd.pvcnn import PVCNN2_PC2
    g = load("pc2_full_n1024.npz")
    net = build(PVCNN2_PC2, g, extra_feature_channels=387)
    x = point_cloud_inputs(1, 390, 1024, int(g["input_seed"]))
    y = net(x.cuda(), torch.from_numpy(g["t"]).cuda()).cpu()
    check_rel_l2(y, torch.from_numpy(g["out"]), TOL)


def test_pvd_full_golden(hip):
    from bdm_amd.pvcnn import PVCNN2_PVD
    g = load("pvd_full_n1024.npz")
    net = build(PVCNN2_PVD, g, extra_feature_channels=0)
    x = point_cloud_inputs(2, 3, 1024, int(g["input_seed"]))
    y = net(x.cuda(), torch.from_numpy(g["t"]).cuda()).cpu()
    check_rel_l2(y, torch.from_numpy(g["out"]), TOL)


def test_fuse_full_golden(hip):
    from bdm_amd.pvcnn import PVCNN2_PC2, PVCNN2_PVD, PVCNN_fuse
    from bdm_amd.utils.procedural import fill_module_
    g = load("fuse_full_n1024.npz")

    class NS:
        pass
    pvd_h, pc2_h = NS(), NS()
    pvd_h.model = NS(); pvd_h.model.module = PVCNN2_PVD(3, 64, extra_feature_channels=0)
    pc2_h.point_cloud_model = NS(); pc2_h.point_cloud_model.model = PVCNN2_PC2(3, 64, extra_feature_channels=387)
    net = PVCNN_fuse(pvd_h, pc2_h, num_classes=3, embed_dim=64, extra_feature_channels=387).eval()
    fill_module_(net, seed=int(g["weight_seed"]))
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    net = net.cuda()
    xr = point_cloud_inputs(1, 390, 1024, int(g["recon_seed"]))
    xp = point_cloud_inputs(1, 3, 1024, int(g["prior_seed"]))
    y = net(xr.cuda(), xp.cuda(), torch.from_numpy(g["t"]).cuda(), "fusion_nstep").cpu()
    check_rel_l2(y, torch.from_numpy(g["out"]), TOL)


@pytest.mark.parametrize("which", ["pc2", "pvd"])
def test_network_goldens_with_the_compact_first_convolution_on_every_level(hip, monkeypatch, which):
    """The default picks the first convolution's form per layer from (batch, points, resolution, channels) (ops.sparse_dil_pays: the
    compact output-stationary kernel where a batch fills the chip with tiles, GEMM + gather elsewhere); the goldens run at B <= 2,
    i.e. on GEMM + gather.  Here the compact form (dilated plan -> sparse_conv_dil -> compact operand split) is forced on EVERY PVConv
    of both denoisers, 8^3 levels included, against the reference's own outputs."""
    from bdm_amd.modules import PVConv
    from bdm_amd.pvcnn import PVCNN2_PC2, PVCNN2_PVD
    from bdm_amd import ops as ops_mod
    monkeypatch.setattr(PVConv, "sparse_dil_always", True)
    monkeypatch.setattr(PVConv, "sparse_dil_resolutions", {8, 16, 32})
    # ... and the rest of the voxel branch on voxel lists (second convolution on the twice-dilated list, SE gate and devoxelisation
    # from its rows + 27 class constants) wherever the module has no attention block
    monkeypatch.setattr(PVConv, "compact_tail", "always")
    monkeypatch.setattr(PVConv, "compact_tail_resolutions", {8, 16, 32})
    tails = []
    real_tail = ops_mod.second_conv_rows
    monkeypatch.setattr(ops_mod, "second_conv_rows", lambda *a, **k: (tails.append(a[4].r), real_tail(*a, **k))[1])
    seen = []
    from bdm_amd import ops
    real = ops.sparse_first_conv_os
    monkeypatch.setattr(ops, "sparse_first_conv_os", lambda *a, **k: (seen.append(a[1].r), real(*a, **k))[1])
    if which == "pc2":
        g = load("pc2_full_n1024.npz")
        net = build(PVCNN2_PC2, g, extra_feature_channels=387)
        x = point_cloud_inputs(1, 390, 1024, int(g["input_seed"]))
    else:
        g = load("pvd_full_n1024.npz")
        net = build(PVCNN2_PVD, g, extra_feature_channels=0)
        x = point_cloud_inputs(2, 3, 1024, int(g["input_seed"]))
    y = net(x.cuda(), torch.from_numpy(g["t"]).cuda()).cpu()
    check_rel_l2(y, torch.from_numpy(g["out"]), TOL)
    assert sorted(set(seen)) == [8, 16, 32] and len(seen) >= 13   # (PC^2: SA0.0 takes the hoisted map)
    assert sorted(set(tails)) == [8, 16, 32] and len(tails) >= 12  # every PVConv but the ones with an attention block


@pytest.mark.parametrize("N,B", [(4096, 2), (2048, 1), (8192, 1), (1100, 3)])
def test_pvd_vs_oracle_fresh_inputs(hip, oracle_ops, N, B):
    """level-0 sizes of the benchmark (N = 4096) against the oracle."""
    from bdm_amd.pvcnn import PVCNN2_PVD
    from bdm_amd.utils.procedural import fill_module_
    from oracle import ref_net
    net = fill_module_(PVCNN2_PVD(3, 64, extra_feature_channels=0).eval(), seed=7)
    x = point_cloud_inputs(B, 3, N, seed=100 + N)
    t = torch.tensor([250] * B)
    ref = ref_net.pvcnn_forward(net.state_dict(), x, t)
    got = net.cuda()(x.cuda(), t.cuda()).cpu()
    check_rel_l2(got, ref, TOL)


@pytest.mark.parametrize("N,B", [(4096, 1), (1500, 2)])
def test_pc2_vs_oracle_fresh_inputs(hip, oracle_ops, N, B):
    """the 390-channel PC^2 denoiser at benchmark size and at a ragged point count, against the oracle."""
    from bdm_amd.pvcnn import PVCNN2_PC2
    from bdm_amd.utils.procedural import fill_module_
    from oracle import ref_net
    net = fill_module_(PVCNN2_PC2(3, 64, extra_feature_channels=387).eval(), seed=11)
    x = point_cloud_inputs(B, 390, N, seed=200 + N)
    t = torch.tensor([[730], [5, 999]][B - 1])
    ref = ref_net.pvcnn_forward(net.state_dict(), x, t)
    got = net.cuda()(x.cuda(), t.cuda()).cpu()
    check_rel_l2(got, ref, TOL)


@pytest.mark.parametrize("conv,sparse_gemm,attention,point_stream", [
    ("fp16x3", "sparse_s3", "bf16x6", True), ("bf16x6", "sparse_s3", "bf16x6", False), ("bf16x6", "sparse", "bf16x6", True),
    pytest.param("fp32", "sparse", "fp32", True, marks=experimental), pytest.param("bf16x6", "sparse", "fp32", False, marks=experimental)])
def test_arithmetic_and_stream_modes_all_match_the_golden(hip, monkeypatch, conv, sparse_gemm, attention, point_stream):
    """every selectable kernel family (BDM_CONV / BDM_SPARSE_GEMM / BDM_ATTENTION / PVConv.point_stream) reproduces the reference."""
    from bdm_amd import ops
    from bdm_amd.modules import PVConv
    from bdm_amd.pvcnn import PVCNN2_PVD
    monkeypatch.setattr(PVConv, "conv_impl", conv)
    monkeypatch.setattr(PVConv, "sparse_gemm", sparse_gemm)
    monkeypatch.setattr(PVConv, "point_stream", point_stream)
    monkeypatch.setattr(ops, "ATTENTION_IMPL", attention)
    g = load("pvd_full_n1024.npz")
    net = build(PVCNN2_PVD, g, extra_feature_channels=0)
    x = point_cloud_inputs(2, 3, 1024, int(g["input_seed"]))
    y = net(x.cuda(), torch.from_numpy(g["t"]).cuda()).cpu()
    check_rel_l2(y, torch.from_numpy(g["out"]), TOL)


def test_forward_is_run_to_run_deterministic(hip):
    """No float atomics anywhere on the path (the reference's voxeliser has them): two forwards give identical bits, also
    with the sampler chain and the point branches running on their own streams."""
    from bdm_amd.pvcnn import PVCNN2_PC2
    from bdm_amd.utils.procedural import fill_module_
    net = fill_module_(PVCNN2_PC2(3, 64, extra_feature_channels=387).eval(), seed=2).cuda()
    x = point_cloud_inputs(3, 390, 2048, seed=77).cuda()
    t = torch.tensor([10, 500, 990]).cuda()
    a = net(x, t).clone()
    for _ in range(3):
        assert torch.equal(net(x, t), a)


def test_geometry_on_the_sampler_stream_changes_no_bit(hip, monkeypatch):
    """Everything that depends on coordinates only runs ahead on the sampler's stream (FPS / ball query of all levels, the next
    levels' voxel plans, the FP modules' 3-NN searches: pvcnn.plan_sampling_chain).  Inline on the main stream, or with the 3-NN
    searches left to the FP modules, the forward gives the same bits."""
    from bdm_amd import pvcnn
    from bdm_amd.pvcnn import PVCNN2_PC2
    from bdm_amd.utils.procedural import fill_module_
    net = fill_module_(PVCNN2_PC2(3, 64, extra_feature_channels=387).eval(), seed=4).cuda()
    x = point_cloud_inputs(3, 390, 4096, seed=78).cuda()
    t = torch.tensor([20, 400, 980]).cuda()
    ahead = net(x, t).clone()
    assert not pvcnn.NN_PLANS, "every planned 3-NN search is consumed by its FP module"
    monkeypatch.setattr(pvcnn, "SIDE_NN", False)
    assert torch.equal(net(x, t), ahead)
    monkeypatch.setattr(pvcnn, "SIDE_STREAM", False)
    assert torch.equal(net(x, t), ahead)


def test_hoisted_projection_conditioning_equals_the_generic_path(hip, monkeypatch):
    """ops.Conditioning: x_in = [xyz, F[pix]] lets the first linear maps of the PC^2 denoiser (SA0 point branch, first sparse
    convolution, last FP module's first layer) be applied to the conditioning IMAGE once and gathered per step.  Same forward as the
    generic path up to the reassociation of those dot products; points without a pixel, a weight update and a new image are covered."""
    from bdm_amd import _lib as L, ops
    from bdm_amd.pvcnn import PVCNN2_PC2
    from bdm_amd.utils.procedural import fill_module_
    B, N, H, C = 2, 1024, 48, 387
    g = torch.Generator().manual_seed(4)
    net = fill_module_(PVCNN2_PC2(3, 64, extra_feature_channels=C).eval(), seed=6).cuda()
    t = torch.tensor([700, 20]).cuda()

    def inputs(seed):
        gg = torch.Generator().manual_seed(seed)
        feat = torch.randn(B, H * H, C, generator=gg).cuda()
        x_t = (torch.randn(B, N, 3, generator=gg) * 0.5).cuda()
        pix = torch.randint(-1, H * H, (B, N), generator=gg, dtype=torch.int32).cuda()
        pix[0, :100] = -1                                                    # points that own no pixel
        x_cf = torch.empty(B, 3 + C, N, device="cuda")
        L.check(L.lib().bdm_condition_gather_cf(B, N, C, H * H, L.ptr(x_t), L.ptr(feat), L.ptr(pix), L.ptr(x_cf), L.stream()), "gather")
        return feat, x_t, pix, x_cf

    maps = {}
    feat, x_t, pix, x_cf = inputs(1)
    ref = net(x_cf, t).clone()                                               # no handle: generic path
    x_cf._bdm_cond = ops.Conditioning(feat, (H, H), pix, x_t, x_cf, maps)
    got = net(x_cf, t)
    assert len(maps) >= 3 and rel_l2(got.cpu(), ref.cpu()) < 5e-6, rel_l2(got.cpu(), ref.cpu())
    assert not torch.equal(got, ref)                                         # really another route
    monkeypatch.setattr(ops, "HOIST_CONDITIONING", False)
    assert torch.equal(net(x_cf, t), ref)                                    # switch off: the generic path, bit for bit
    monkeypatch.setattr(ops, "HOIST_CONDITIONING", True)
    # a second step on the same image re-uses the maps; other points, other pixels
    n_maps = {k: v[1].data_ptr() for k, v in maps.items() if torch.is_tensor(v[1])}
    _, x_t2, pix2, x_cf2 = inputs(2)
    L.check(L.lib().bdm_condition_gather_cf(B, N, C, H * H, L.ptr(x_t2), L.ptr(feat), L.ptr(pix2), L.ptr(x_cf2), L.stream()), "gather")
    ref2 = net(x_cf2, t).clone()
    x_cf2._bdm_cond = ops.Conditioning(feat, (H, H), pix2, x_t2, x_cf2, maps)
    assert rel_l2(net(x_cf2, t).cpu(), ref2.cpu()) < 5e-6
    assert {k: v[1].data_ptr() for k, v in maps.items() if torch.is_tensor(v[1])} == n_maps
    # rewriting a weight invalidates its map
    with torch.no_grad():
        net.sa_layers[0][0].point_features.layers[0].weight.mul_(1.5)
    x_cf2._bdm_cond = None
    ref3 = net(x_cf2, t).clone()
    x_cf2._bdm_cond = ops.Conditioning(feat, (H, H), pix2, x_t2, x_cf2, maps)
    assert rel_l2(net(x_cf2, t).cpu(), ref3.cpu()) < 5e-6 and rel_l2(ref3.cpu(), ref2.cpu()) > 1e-4


@pytest.mark.parametrize("which", ["pc2", "pvd"])
def test_time_embedding_as_a_per_shape_bias_of_the_fp_modules(hip, monkeypatch, which):
    """pvcnn.decode in split form: the point-invariant time embedding is not concatenated to the features before a PointNetFPModule
    (pointnet.py:104-112 interpolates cat([features, t_emb]) AND t_emb: both are t itself); its share of the first MLP layer enters as
    a per-shape bias W[:, t columns] . t (all four modules' in one launch).  Same forward up to the reassociation of that share."""
    import bdm_amd.pvcnn as PV
    from bdm_amd.utils.procedural import fill_module_
    B, N = 3, 2048
    net = fill_module_((PV.PVCNN2_PC2(3, 64, extra_feature_channels=32) if which == "pc2" else PV.PVCNN2_PVD(3, 64, extra_feature_channels=0)).eval(),
                       seed=9).cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(B, 3, N, generator=g) * 0.4, torch.randn(B, 32 if which == "pc2" else 0, N, generator=g)], dim=1).cuda()
    t = torch.tensor([900, 400, 3]).cuda()
    monkeypatch.setattr(PV, "FP_TEMB_SPLIT", False)
    ref = net(x, t).clone()
    monkeypatch.setattr(PV, "FP_TEMB_SPLIT", True)
    got = net(x, t)
    check_rel_l2(got.cpu(), ref.cpu(), 2e-6)
    assert not torch.equal(got, ref)                       # really another route
    assert torch.equal(net(x, t), got)                     # deterministic
    alone = net(x[1:2].contiguous(), t[1:2])
    assert torch.equal(alone[0], got[1])                   # and batch-invariant


@pytest.mark.parametrize("which", ["pc2", "pvd"])
def test_time_embedding_as_per_shape_terms_of_the_encoder(hip, monkeypatch, which):
    """pvcnn.encode in split form: the point-invariant time embedding is not concatenated before the first PVConv of levels 1, 2 and before
    the last set-abstraction module (pvcnn.py:103).  Its voxel mean is the same constant on every occupied cell (vox.cu:18-72), so its share of
    the first voxel convolution is a per-shape column addend of the occupied-row GEMM, added by the gather once per occupied neighbour
    (bdm_sparse_conv_gemm_*_cb) = the zero-padded convolution of the concatenated grid; its share of the point branch and of the grouped
    MLP is a per-shape bias.  Same forward up to the reassociation of those shares (and the activation scale of a narrower operand)."""
    import bdm_amd.pvcnn as PV
    from bdm_amd.modules import PVConv, PointNetSAModule
    from bdm_amd.utils.procedural import fill_module_
    B, N = 3, 2048
    net = fill_module_((PV.PVCNN2_PC2(3, 64, extra_feature_channels=32) if which == "pc2" else PV.PVCNN2_PVD(3, 64, extra_feature_channels=0)).eval(),
                       seed=9).cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(B, 3, N, generator=g) * 0.4, torch.randn(B, 32 if which == "pc2" else 0, N, generator=g)], dim=1).cuda()
    t = torch.tensor([900, 400, 3]).cuda()
    took = []
    for cls in (PVConv, PointNetSAModule):
        orig = cls.can_split_temb
        monkeypatch.setattr(cls, "can_split_temb", lambda self, f, te, orig=orig: (took.append(type(self).__name__ if orig(self, f, te) else None) or took[-1] is not None))
    monkeypatch.setattr(PVConv, "temb_split", False)
    ref = net(x, t).clone()
    assert not any(took)
    monkeypatch.setattr(PVConv, "temb_split", True)
    del took[:]
    got = net(x, t)
    assert [k for k in took if k] .count("PVConv") >= 2 and "PointNetSAModule" in took, took    # levels 1, 2 and the last module
    check_rel_l2(got.cpu(), ref.cpu(), 5e-6)
    assert not torch.equal(got, ref)                       # really another route
    assert torch.equal(net(x, t), got)                     # deterministic
    alone = net(x[1:2].contiguous(), t[1:2])
    assert torch.equal(alone[0], got[1])                   # and batch-invariant


def test_decoder_voxel_plan_on_the_sampler_stream_same_bits(hip, monkeypatch):
    """pvcnn.plan_sampling_chain also builds the voxel plan of the FP stage no encoder level shares (FP0's 64 points) on the sampler's
    stream: the forward's bits do not change, and neither do they when a forward that aborted midway left per-shape time-embedding terms behind."""
    import bdm_amd.pvcnn as PV
    from bdm_amd.utils.procedural import fill_module_
    B, N = 2, 4096
    net = fill_module_(PV.PVCNN2_PC2(3, 64, extra_feature_channels=32).eval(), seed=9).cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(B, 3, N, generator=g) * 0.4, torch.randn(B, 32, N, generator=g)], dim=1).cuda()
    t = torch.tensor([900, 3]).cuda()
    monkeypatch.setattr(PV, "DECODER_PLAN", False)
    ref = net(x, t).clone()
    monkeypatch.setattr(PV, "DECODER_PLAN", True)
    assert torch.equal(net(x, t), ref)
    # stale terms of ANOTHER timestep on every encoder module that takes them: encode must not use them
    net(x, torch.tensor([17, 512]).cuda())
    for blk in list(net.sa_layers)[1:]:
        first = blk[0] if isinstance(blk, torch.nn.Sequential) else blk
        if hasattr(first, "temb_rows"):
            rows = first.temb_rows(64).shape[0]
            first._temb_terms = torch.full((B, rows), 7.0, device="cuda")
    assert torch.equal(net(x, t), ref)


def test_pvconv_takes_the_time_embedding_as_column_addend_and_bias(hip, monkeypatch):
    """One PVConv of each encoder kind (16^3: bf16x6 GEMM; 8^3: fp16x3 GEMM) against itself on the concatenated input, plus the refusal of
    an input of the wrong width."""
    from bdm_amd import ops
    from bdm_amd.modules import PVConv
    from bdm_amd.utils.procedural import fill_module_
    g = torch.Generator().manual_seed(5)
    for (cf, cout, r, n) in ((64, 64, 16, 1024), (128, 128, 8, 256)):
        B, ct = 3, 64
        m = fill_module_(PVConv(cf + ct, cout, 3, r, with_se=True, with_se_relu=True).eval(), seed=4).cuda()
        f = torch.randn(B, cf, n, generator=g).cuda()
        coords = (torch.randn(B, 3, n, generator=g) * 0.4).cuda()
        te = torch.randn(B, ct, generator=g).cuda()[:, :, None].expand(-1, -1, n)
        ref = m((torch.cat([f, te], dim=1).contiguous(), coords, te))[0].clone()
        ops.clear_plan_cache()
        got = m((f, coords, te))[0]
        check_rel_l2(got.cpu(), ref.cpu(), 3e-6, f"PVConv {cf}+{ct} -> {cout} at {r}^3")
        assert not torch.equal(got, ref)
        ops.clear_plan_cache()
        with pytest.raises(ValueError):
            m((f[:, :cf - 8].contiguous(), coords, te))
        monkeypatch.setattr(PVConv, "temb_split", False)
        ops.clear_plan_cache()
        with pytest.raises(ValueError):
            m((f, coords, te))
        monkeypatch.setattr(PVConv, "temb_split", True)
        ops.clear_plan_cache()
