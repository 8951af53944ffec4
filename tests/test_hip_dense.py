"""GPU parity of the dense C-ABI operators vs plain PyTorch fp32 on the CPU (tolerances per test)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ops(hip):
    from bdm_amd import ops as o
    return o


def _has_experimental():
    from bdm_amd import _lib
    return _lib.has_experimental()


from helpers import experimental  # noqa: E402  (tests of the EXPERIMENTAL=1 kernel families)


@pytest.mark.parametrize("B,M,K,N", [(2, 64, 35, 1024 * 32), (2, 32, 390, 4096), (1, 3, 128, 1100), (3, 512, 323, 16),
                                     (2, 256, 832, 64), (1, 192, 64, 4096), (2, 8, 8, 33)])
def test_pointwise_conv(ops, B, M, K, N):
    g = torch.Generator().manual_seed(M * K)
    x = torch.randn(B, K, N, generator=g)
    w = torch.randn(M, K, 1, generator=g) / K ** 0.5
    b = torch.randn(M, generator=g)
    ref = TF.conv1d(x.double(), w.double(), b.double()).float()
    got = ops.pointwise_conv(x.cuda(), w.cuda(), b.cuda()).cpu()
    assert rel(got, ref) < 2e-6  # fp32 MFMA, k-ordered accumulation vs fp64 reference


@experimental
@pytest.mark.parametrize("B,M,K,N", [(2, 64, 35, 1024 * 32), (2, 32, 390, 4096), (1, 3, 128, 1100), (1, 192, 64, 4096), (2, 8, 8, 133),
                                     (2, 100, 579, 1000), (3, 512, 256, 512), (2, 256, 323, 65)])
def test_pointwise_conv_bf16x6_and_fp32_kernels(ops, monkeypatch, B, M, K, N):
    """The two kernel families of the 1x1 convolutions (experimental/pointwise_s3.hip: exact bf16 triples, six products, opt-in;
    dense_ops.hip: fp32-input MFMA) against fp64, with every epilogue term, a strided operand and K / M / N tails; and the
    GroupNorm statistics the bf16x6 kernel leaves against the values it wrote."""
    g = torch.Generator().manual_seed(M * K + N)
    big = torch.randn(B, K + 3, N, generator=g).cuda()
    x = big[:, 1:1 + K]
    w = (torch.randn(M, K, generator=g) / K ** 0.5).cuda()
    bias, bb, res = torch.randn(M, generator=g).cuda(), torch.randn(B, M, generator=g).cuda(), torch.randn(B, M, N, generator=g).cuda()
    lin = TF.conv1d(x.cpu().double(), w.cpu().double()[:, :, None], bias.cpu().double()) + bb.cpu().double()[:, :, None]
    ref = TF.leaky_relu(lin, 0.1) + res.cpu().double()
    outs = {}
    for impl in ("bf16x6", "fp32"):
        monkeypatch.setattr(ops, "PW_IMPL", impl)
        got = ops.pointwise_conv(x, w, bias, batch_bias=bb, act=2, slope=0.1, residual=res)
        assert rel(got.cpu(), ref.float()) < 2e-6, impl
        assert torch.equal(got, ops.pointwise_conv(x, w, bias, batch_bias=bb, act=2, slope=0.1, residual=res)), impl  # deterministic
        outs[impl] = got
    assert not torch.equal(outs["bf16x6"], outs["fp32"])  # really two different kernels
    if M % 8 == 0 and ops.gn_foldable(M, 8):
        monkeypatch.setattr(ops, "PW_IMPL", "bf16x6")
        y, (partial, slices, groups) = ops.pointwise_conv_gn(x, w, bias, out_groups=8)
        assert rel(y.cpu(), (lin - bb.cpu().double()[:, :, None]).float()) < 2e-6
        sums = partial.view(B, 8, slices, 2).sum(2)
        yd = y.double().view(B, 8, -1)
        assert torch.allclose(sums[..., 0], yd.sum(-1), rtol=1e-6, atol=1e-3) and torch.allclose(sums[..., 1], (yd * yd).sum(-1), rtol=1e-6)


def test_pointwise_conv_views_residual_leaky(ops):
    g = torch.Generator().manual_seed(5)
    B, K, M, N = 2, 67, 40, 300
    big = torch.randn(B, K + 9, N, generator=g).cuda()
    x = big[:, 4:4 + K]
    w = (torch.randn(M, K, generator=g) / 8).cuda()
    bias = torch.randn(M, generator=g).cuda()
    res = torch.randn(B, M, N, generator=g).cuda()
    outbuf = torch.zeros(B, M + 5, N).cuda()
    ops.pointwise_conv(x, w, bias, out=outbuf[:, 5:], act=2, slope=0.02, residual=res)
    ref = TF.leaky_relu(TF.conv1d(x.cpu(), w.cpu()[:, :, None], bias.cpu()), 0.02) + res.cpu()
    assert rel(outbuf[:, 5:].cpu(), ref) < 2e-6
    assert float(outbuf[:, :5].abs().max()) == 0.0


@pytest.mark.parametrize("B,M,K,N", [(16, 1536, 512, 16), (3, 512, 512, 16), (2, 100, 136, 32), (2, 40, 131, 7), (1, 33, 1000, 1), (2, 64, 128, 20)])
def test_pointwise_conv_skinny(ops, B, M, K, N):
    """n <= 32 columns and k >= 128 (the 16-point level): pw_skinny_kernel, K split over the four waves and summed in wave order.
    Aligned and unaligned weights (the 16-byte A loads need ldw % 4 == 0), K tails, row tails, strided views, every epilogue term."""
    g = torch.Generator().manual_seed(M * K + N)
    big = torch.randn(B, K + 3, N, generator=g).cuda()
    x = big[:, 2:2 + K]                                    # batch stride != K * N
    w = (torch.randn(M, K, generator=g) / K ** 0.5).cuda()
    bias, bb = torch.randn(M, generator=g).cuda(), torch.randn(B, M, generator=g).cuda()
    res = torch.randn(B, M, N, generator=g).cuda()
    lin = TF.conv1d(x.cpu().double(), w.cpu().double()[:, :, None], bias.cpu().double()) + bb.cpu().double()[:, :, None]
    got = ops.pointwise_conv(x, w, bias, batch_bias=bb)
    assert rel(got.cpu(), lin.float()) < 2e-6
    assert torch.equal(got, ops.pointwise_conv(x, w, bias, batch_bias=bb))   # deterministic
    out = torch.zeros(B, M + 2, N).cuda()
    ops.pointwise_conv(x, w, bias, batch_bias=bb, out=out[:, 2:], act=2, slope=0.1, residual=res)
    ref = TF.leaky_relu(lin, 0.1) + res.cpu().double()
    assert rel(out[:, 2:].cpu(), ref.float()) < 2e-6 and float(out[:, :2].abs().max()) == 0.0


@pytest.mark.parametrize("B,C,L", [(2, 64, 32768), (3, 32, 1024 * 32), (2, 512, 16), (1, 8, 33), (2, 256, 512)])
@pytest.mark.parametrize("swish", [False, True])
def test_group_norm(ops, B, C, L, swish):
    g = torch.Generator().manual_seed(C + L)
    x = torch.randn(B, C, L, generator=g) * 3 + 1
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = TF.group_norm(x, 8, ga, be, 1e-5)
    if swish:
        ref = ref * torch.sigmoid(ref)
    got = ops.group_norm_(x.cuda(), ga.cuda(), be.cuda(), 8, 1e-5, swish=swish).cpu()
    assert rel(got, ref) < 3e-6


def test_group_norm_residual(ops):
    g = torch.Generator().manual_seed(3)
    x, r = torch.randn(2, 64, 4096, generator=g), torch.randn(2, 64, 4096, generator=g)
    ga, be = torch.randn(64, generator=g), torch.randn(64, generator=g)
    ref = TF.group_norm(x + r, 8, ga, be)
    ref = ref * torch.sigmoid(ref)
    got = ops.group_norm_(x.cuda(), ga.cuda(), be.cuda(), swish=True, residual=r.cuda()).cpu()
    assert rel(got, ref) < 3e-6


@experimental
@pytest.mark.parametrize("cin,cout,r", [(35, 32, 32), (32, 32, 32), (64, 64, 32), (128, 64, 16), (128, 128, 16),
                                        (192, 128, 8), (256, 256, 8), (7, 8, 8), (16, 40, 16)])
def test_conv3d(ops, cin, cout, r):
    B = 2
    g = torch.Generator().manual_seed(cin * cout + r)
    x = torch.randn(B, cin, r, r, r, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = TF.conv3d(x.double(), w.double(), b.double(), padding=1).float().reshape(B, cout, -1)
    pw = ops.conv3d_pack(w.cuda())
    got = ops.conv3d(x.cuda().reshape(B, cin, -1), pw, b.cuda(), r).cpu()
    assert rel(got, ref) < 2e-6


@experimental
def test_conv3d_boundary_exact(ops):
    """all-ones input and weights: interior = 27*cin, faces/edges/corners smaller (zero padding)."""
    cin, cout, r = 8, 32, 8
    x = torch.ones(1, cin, r ** 3).cuda()
    w = torch.ones(cout, cin, 3, 3, 3).cuda()
    y = ops.conv3d(x, ops.conv3d_pack(w), None, r).cpu().view(cout, r, r, r)
    assert float(y[0, 3, 3, 3]) == 27 * cin and float(y[5, 0, 0, 0]) == 8 * cin
    assert float(y[31, 0, 4, 4]) == 18 * cin and float(y[7, 0, 0, 4]) == 12 * cin


@pytest.mark.parametrize("B,C,L", [(2, 64, 4096), (1, 16, 4096), (2, 512, 16), (1, 128, 16), (1, 64, 512), (1, 24, 200),
                                   (3, 64, 197), (2, 40, 1000)])
def test_attention_core(ops, B, C, L):
    g = torch.Generator().manual_seed(C * L)
    qkv = torch.randn(B, 3 * C, L, generator=g) * (0.5 if L > 64 else 0.2)
    q, k, v = qkv[:, :C].double(), qkv[:, C:2 * C].double(), qkv[:, 2 * C:].double()
    w = torch.softmax(torch.matmul(q.permute(0, 2, 1), k), -1)
    ref = torch.matmul(v, w.permute(0, 2, 1)).float()
    for impl in ("bf16x6", "fp32") if (_has_experimental() or L <= 64) else ("bf16x6",):
        got = ops.attention_core(qkv.cuda(), C, impl=impl).cpu()
        assert rel(got, ref) < 5e-6, impl
    if L > 64 and C <= 64:  # fp16x3 kernel: scales from max |q|, |k|, |v| (here computed on the host; the projection GEMM leaves them)
        for scale in (1.0, 1e-4, 300.0):   # tensors far below / above fp16's comfortable range: the scales absorb it
            qs = qkv.clone()
            qs[:, 2 * C:] *= scale         # v
            qs[:, :C] *= min(scale, 4.0)   # q (a large q.k would make the softmax one-hot: keep it moderate)
            qd, kd, vd = qs[:, :C].double(), qs[:, C:2 * C].double(), qs[:, 2 * C:].double()
            refs = torch.matmul(vd, torch.softmax(torch.matmul(qd.permute(0, 2, 1), kd), -1).permute(0, 2, 1)).float()
            amax = torch.stack([qs[:, i * C:(i + 1) * C].abs().amax(dim=(1, 2)) for i in range(3)], 1).float().contiguous().cuda()  # (B, 3)
            got = ops.attention_core(qs.cuda(), C, amax=amax).cpu()
            assert rel(got, refs) < 5e-6, ("fp16x3", scale)


@pytest.mark.parametrize("B,C,L", [(1, 64, 4096), (2, 32, 4100), (5, 64, 1500)])
def test_attention_key_ranges_equal_the_single_range_form(ops, monkeypatch, B, C, L):
    """bdm_attention_core_h2 splits a query's keys into 1 / 2 / 4 ranges by (shapes, positions) (bdm_attention_h2_key_slices) and merges the
    partial results (attn_combine_kernel): every count against the float64 softmax, against the single-range form, deterministic, and
    the default equal to the count the library reports."""
    from bdm_amd import _lib as L_
    g = torch.Generator().manual_seed(B * L)
    qkv = torch.randn(B, 3 * C, L, generator=g) * 0.6
    q, k, v = qkv[:, :C].double(), qkv[:, C:2 * C].double(), qkv[:, 2 * C:].double()
    ref = torch.matmul(v, torch.softmax(torch.matmul(q.permute(0, 2, 1), k), -1).permute(0, 2, 1)).float()
    amax = torch.stack([qkv[:, i * C:(i + 1) * C].abs().amax(dim=(1, 2)) for i in range(3)], 1).float().contiguous().cuda()
    x = qkv.cuda()
    outs = {}
    for ks in (1, 2, 3, 4, 8):
        monkeypatch.setattr(ops, "ATTN_KSPLIT", ks)
        outs[ks] = ops.attention_core(x, C, amax=amax).cpu()
        assert rel(outs[ks], ref) < 5e-6, ks
        assert torch.equal(outs[ks], ops.attention_core(x, C, amax=amax).cpu()), ks   # deterministic
        if ks > 1:
            assert rel(outs[ks], outs[1]) < 2e-6 and not torch.equal(outs[ks], outs[1]), ks
    monkeypatch.setattr(ops, "ATTN_KSPLIT", None)
    auto = L_.lib().bdm_attention_h2_key_slices(B, L)
    assert auto in (1, 2, 4) and torch.equal(ops.attention_core(x, C, amax=amax).cpu(), outs[auto])


def test_projection_gemm_leaves_operand_maxima_for_the_fp16x3_attention(ops):
    """bdm_pointwise_conv_gn amax output: max |y| per (shape, block of amax_rows rows), and the Attention module on the fp16x3
    path vs the bf16x6 path."""
    g = torch.Generator().manual_seed(3)
    B, C, L = 3, 64, 1000
    x = torch.randn(B, C, L, generator=g).cuda()
    w, b = (torch.randn(3 * C, C, generator=g) / 8).cuda(), torch.randn(3 * C, generator=g).cuda()
    amax = ops.amax_slots(x.device, 3 * B)
    y = ops.pointwise_conv_gn(x, w, b, amax=amax, amax_rows=C)
    assert torch.equal(y, ops.pointwise_conv(x, w, b))
    want = torch.stack([y[:, i * C:(i + 1) * C].abs().amax(dim=(1, 2)) for i in range(3)], 1)
    assert torch.equal(amax.view(B, 3), want)
    from bdm_amd.modules import Attention
    from bdm_amd.utils.procedural import fill_module_
    att = fill_module_(Attention(64, 8, D=3).eval(), seed=2).cuda()
    v = torch.randn(2, 64, 16, 16, 16, generator=g).cuda()
    import bdm_amd.ops as O
    old_impl = O.ATTENTION_IMPL
    try:
        O.ATTENTION_IMPL = "bf16x6"
        ref = att(v).clone()
        O.ATTENTION_IMPL = "fp16x3"
        got = att(v)
    finally:
        O.ATTENTION_IMPL = old_impl
    assert rel(got.cpu(), ref.cpu()) < 2e-6


def test_small_ops(ops):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 70, 130, 32, generator=g)
    assert torch.equal(ops.max_over_neighbors(x.cuda()).cpu(), x.max(-1).values)
    # SE gate
    C, r = 64, 16
    v = torch.randn(2, C, r ** 3, generator=g)
    w1, w2 = torch.randn(C // 8, C, generator=g) / 8, torch.randn(C, C // 8, generator=g) / 3
    ref = torch.sigmoid(TF.linear(torch.relu(TF.linear(v.mean(-1), w1)), w2))
    two = ops.se_gate(v.cuda(), w1.cuda(), w2.cuda(), fused=False).cpu()
    assert rel(two, ref) < 2e-6
    # one-launch form (last workgroup of a shape runs the FC layers): identical bits, repeatedly, on big batches too
    for _ in range(20):
        assert torch.equal(ops.se_gate(v.cuda(), w1.cuda(), w2.cuda(), fused=True).cpu(), two)
    vb = torch.randn(16, 256, 512, generator=g).cuda()
    w1b, w2b = (torch.randn(32, 256, generator=g) / 16).cuda(), (torch.randn(256, 32, generator=g) / 6).cuda()
    twob = ops.se_gate(vb, w1b, w2b, fused=False)
    for _ in range(10):
        assert torch.equal(ops.se_gate(vb, w1b, w2b, fused=True), twob)
    # time embedding
    from oracle import ref_net
    sd = {"0.weight": torch.randn(64, 64, generator=g) / 8, "0.bias": torch.randn(64, generator=g),
          "2.weight": torch.randn(64, 64, generator=g) / 8, "2.bias": torch.randn(64, generator=g)}
    t = torch.tensor([0, 1, 17, 500, 999])
    ref = ref_net.embedf(sd, "", t, 64)
    got = ops.time_embedding(t.cuda(), *(sd[k].cuda() for k in ("0.weight", "0.bias", "2.weight", "2.bias"))).cpu()
    assert rel(got, ref) < 2e-5  # sin/cos of arguments up to 999 rad: device vs host libm
    # transpose, concat, broadcast
    a = torch.randn(2, 100, 37, generator=g)
    assert torch.equal(ops.transpose12(a.cuda()).cpu(), a.transpose(1, 2).contiguous())
    te = torch.randn(2, 64, generator=g).cuda()
    f = torch.randn(2, 10, 77, generator=g).cuda()
    cat = ops.cat_channels([f, te[:, :, None].expand(-1, -1, 77), f[:, 2:5]])
    assert torch.equal(cat.cpu(), torch.cat([f, te[:, :, None].expand(-1, -1, 77), f[:, 2:5]], 1).cpu())


@pytest.mark.parametrize("n,r", [(4096, 32), (1024, 16), (64, 8), (1100, 32)])
def test_voxel_coords(ops, n, r):
    from oracle import ref_net
    g = torch.Generator().manual_seed(n)
    coords = torch.randn(2, 3, n, generator=g) * 0.4 + 0.1
    ref_nc, ref_vc = ref_net.voxel_coords(coords, r)
    nc, vc = ops.voxel_coords(coords.cuda(), r)
    assert float((nc.cpu() - ref_nc).abs().max()) < 1e-4 * r  # reductions differ in the last ulp
    # rounding may flip only where the coordinate sits within float noise of .5
    flips = (vc.cpu() != ref_vc)
    assert int(flips.sum()) <= 2
    assert bool(((ref_nc - ref_nc.floor() - 0.5).abs()[flips] < 1e-3).all())


def test_pvconv_tail_and_sa_group(ops, oracle_ops):
    g = torch.Generator().manual_seed(2)
    B, C, n, r = 2, 32, 1000, 16
    nc = torch.rand(B, 3, n, generator=g) * (r - 1)
    grid = torch.randn(B, C, r ** 3, generator=g)
    gate = torch.rand(B, C, generator=g)
    add = torch.randn(B, C, n, generator=g)
    ref = oracle_ops.trilinear_devoxelize_forward(r, False, nc, (grid * gate[:, :, None]).contiguous())[0] + add
    got = ops.devoxelize_gate_add(nc.cuda(), grid.cuda(), r, gate.cuda(), add.cuda()).cpu()
    assert torch.equal(got, ref)  # tolerance 0: same unfused order
    pts = torch.randn(B, 3, n, generator=g) * 0.3
    idx = oracle_ops.furthest_point_sampling(pts, 50)
    ctr = oracle_ops.gather_features_forward(pts, idx)
    nb = oracle_ops.ball_query(ctr, pts, 0.2, 32)
    f = torch.randn(B, 11, n, generator=g)
    ref = torch.cat([oracle_ops.grouping_forward(pts, nb) - ctr.unsqueeze(-1), oracle_ops.grouping_forward(f, nb)], 1)
    for point_major in (False, True):
        got = ops.sa_group(pts.cuda(), ctr.cuda(), f.cuda(), nb.cuda(), point_major=point_major).cpu()
        assert torch.equal(got, ref)
    big = torch.zeros(B, 300, n)  # channel slice of a wider buffer (strided rows), 259 channels: 9 column blocks
    big[:, 20:279] = torch.randn(B, 259, n, generator=g)
    fv = big[:, 20:279]
    ref = torch.cat([oracle_ops.grouping_forward(pts, nb) - ctr.unsqueeze(-1), oracle_ops.grouping_forward(fv.contiguous(), nb)], 1)
    assert torch.equal(ops.sa_group(pts.cuda(), ctr.cuda(), big.cuda()[:, 20:279], nb.cuda()).cpu(), ref)


@experimental
@pytest.mark.parametrize("cin,cout,r,npts", [(35, 32, 32, 4096), (64, 64, 32, 4096), (128, 64, 16, 1024), (256, 256, 8, 64),
                                             (192, 128, 8, 256), (16, 8, 32, 50)])
def test_conv3d_sparse_input_bit_identical(ops, cin, cout, r, npts):
    """first conv of a PVConv: row-occupancy skipping must not change a single bit."""
    B = 2
    g = torch.Generator().manual_seed(cin + r)
    vc = (torch.randn(B, 3, npts, generator=g) * r / 8 + r / 2).round().clamp(0, r - 1).to(torch.int32)
    vc[1] = vc[1] // 2  # second shape squeezed into one octant: large empty regions
    f = torch.randn(B, cin, npts, generator=g)
    vox, rowocc = ops.avg_voxelize(f.cuda(), vc.cuda(), r, with_row_occupancy=True)
    occ_ref = torch.zeros(B, r * r, dtype=torch.uint8)
    for b in range(B):
        occ_ref[b, (vc[b, 0] * r + vc[b, 1]).long()] = 1
    assert torch.equal(rowocc.cpu(), occ_ref)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    pw = ops.conv3d_pack(w.cuda())
    dense = ops.conv3d(vox, pw, bias.cuda(), r)
    sparse = ops.conv3d(vox, pw, bias.cuda(), r, rowocc=rowocc)
    assert torch.equal(dense, sparse)
    ref = TF.conv3d(vox.cpu().double().view(B, cin, r, r, r), w.double(), bias.double(), padding=1).float().reshape(B, cout, -1)
    assert rel(sparse.cpu(), ref) < 2e-6


@pytest.mark.parametrize("cin,cout,r", [(35, 32, 32), (64, 64, 32), (128, 64, 16), (128, 128, 16), (192, 128, 8),
                                        (256, 256, 8), (7, 8, 8), (390, 32, 32)])
def test_conv3d_bf16x6_has_fp32_accuracy(ops, cin, cout, r):
    """split-bf16 (6 partial products) convolution: error vs fp64 must be at the fp32-MFMA kernel's level."""
    B = 2
    g = torch.Generator().manual_seed(cin * cout + r)
    x = torch.randn(B, cin, r ** 3, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = TF.conv3d(x.double().view(B, cin, r, r, r), w.double(), b.double(), padding=1).float().reshape(B, cout, -1)
    xs = ops.to_s3(x.cuda())
    got = ops.conv3d_s3(xs, ops.conv3d_s3_pack(w.cuda()), b.cuda(), cin, cout, r).cpu()
    e6 = rel(got, ref)
    assert e6 < 2e-6, e6
    if _has_experimental():  # ... and at the level of the fp32-MFMA kernel's own error
        e32 = rel(ops.conv3d(x.cuda(), ops.conv3d_pack(w.cuda()), b.cuda(), r).cpu(), ref)
        assert e6 < 4 * e32 + 1e-7, (e6, e32)


@pytest.mark.parametrize("cin,cout,r,kind", [(35, 32, 32, "normal"), (64, 64, 32, "normal"), (128, 64, 16, "wide"), (128, 128, 16, "normal"),
                                             (192, 128, 8, "tiny"), (256, 256, 8, "normal"), (7, 8, 8, "wide"), (64, 64, 16, "saturate")])
def test_conv3d_fp16x3_has_fp32_accuracy(ops, cin, cout, r, kind):
    """two-term fp16 operands, three partial products, power-of-two scaling: error vs fp64 at the fp32-MFMA kernel's level,
    also for weights spanning 8 decades per layer, activations of very different magnitudes and tiny values."""
    B = 2
    g = torch.Generator().manual_seed(cin * cout + r)
    x = torch.randn(B, cin, r ** 3, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    if kind == "wide":    # per-channel weight scales from 1e-4 to 1e4, activations with a 1e-3 .. 30 spread
        w = w * (10.0 ** torch.linspace(-4, 4, cout)).view(-1, 1, 1, 1, 1)
        x = x * (10.0 ** torch.linspace(-3, 1.5, cin)).view(1, -1, 1)
    elif kind == "tiny":  # everything far below fp16's normal range before scaling
        x, w = x * 1e-6, w * 1e-5
    b = torch.randn(cout, generator=g) * float(w.abs().mean() * x.abs().mean()) * 10
    ref = TF.conv3d(x.double().view(B, cin, r, r, r), w.double(), b.double(), padding=1).float().reshape(B, cout, -1)
    if kind == "saturate":  # |scale * x| > 65504 saturates (documented), it must not produce inf / nan
        x[0, 0, 0] = 1e6
        got = ops.conv3d_h2(ops.to_h2(x.cuda(), scale=16.0), ops.conv3d_h2_pack(w.cuda()), b.cuda(), cin, cout, r).cpu()
        assert bool(torch.isfinite(got).all())
        return
    got = ops.conv3d_h2(ops.to_h2(x.cuda()), ops.conv3d_h2_pack(w.cuda()), b.cuda(), cin, cout, r).cpu()
    # per output channel (the weight scales differ by decades): every channel must be fp32-grade
    e3 = ((got - ref).norm(dim=(0, 2)) / ref.norm(dim=(0, 2))).max().item()
    assert e3 < 2e-6, e3
    if _has_experimental():  # ... and at the level of the fp32-MFMA kernel's own error
        fp32 = ops.conv3d(x.cuda(), ops.conv3d_pack(w.cuda()), b.cuda(), r).cpu()
        e32 = ((fp32 - ref).norm(dim=(0, 2)) / ref.norm(dim=(0, 2))).max().item()
        assert e3 < 4 * e32 + 1e-7, (e3, e32)


@pytest.mark.parametrize("cin,cout,r", [(64, 64, 32), (128, 128, 16), (64, 128, 16), (256, 256, 8)])
def test_conv3d_fp16x3_tile_choice_does_not_change_the_bits(ops, cin, cout, r):
    """The launcher picks smaller tiles when a few shapes cannot fill the chip (conv3d_h2.hip: 32-row / 256-voxel tiles at 16^3
    and 32^3, 128-voxel tiles at 8^3): a shape convolved alone and inside a batch of 10 gives the same bits, and the GroupNorm
    statistics the epilogue leaves agree to fp32 rounding of the per-wave sums (a different number of slices)."""
    B = 10
    g = torch.Generator().manual_seed(cin + cout + r)
    x = torch.randn(B, cin, r ** 3, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    b = torch.randn(cout, generator=g).cuda()
    packed = ops.conv3d_h2_pack(w)
    full, (ws, slices) = ops.conv3d_h2_gn(ops.to_h2(x, scale=16.0), packed, b, cin, cout, r)
    one, (ws1, slices1) = ops.conv3d_h2_gn(ops.to_h2(x[3:4].contiguous(), scale=16.0), packed, b, cin, cout, r)
    assert torch.equal(full[3:4], one)
    assert torch.equal(ops.conv3d_h2(ops.to_h2(x[3:4].contiguous(), scale=16.0), packed, b, cin, cout, r), one)
    st = ws.view(torch.float64)[:B * 8 * slices * 2].view(B, 8, slices, 2).sum(2)
    st1 = ws1.view(torch.float64)[:8 * slices1 * 2].view(1, 8, slices1, 2).sum(2)
    ref = one.double().view(8, -1)
    scale = torch.stack([ref.abs().sum(1), (ref * ref).sum(1)], -1)  # fp32 rounding of per-wave partial sums: relative to sum |v|
    assert bool(((st[3] - st1[0]).abs() <= 1e-7 * scale).all())
    assert bool(((st1[0] - torch.stack([ref.sum(1), (ref * ref).sum(1)], -1)).abs() <= 1e-6 * scale).all())


def test_h2_producer(ops):
    """GroupNorm + Swish -> H2 reconstructs ((hi + lo) / 16) the fp32 values of the fp32 GroupNorm kernel."""
    g = torch.Generator().manual_seed(18)
    B, r = 2, 16
    x = torch.randn(B, 40, r ** 3, generator=g) * 2 + 0.5
    gn = torch.nn.GroupNorm(8, 40)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(40, generator=g)); gn.bias.copy_(torch.randn(40, generator=g))
    ref = ops.group_norm_(x.clone().cuda(), gn.weight.cuda(), gn.bias.cuda(), swish=True).cpu()
    h2, inv = ops.to_h2(x.cuda(), gn.cuda(), swish=True)
    got = (h2.cpu().double().sum(2) * inv).permute(0, 1, 3, 2).reshape(B, -1, r ** 3)[:, :40].float()
    assert rel(got, ref) < 3e-7


def test_s3_producers(ops, oracle_ops):
    """GroupNorm(+Swish)->S3 and voxelise->S3 reconstruct (hi+mid+lo) the fp32 values of the fp32 kernels."""
    g = torch.Generator().manual_seed(8)
    B, C, r, n = 2, 35, 16, 1000

    def unsplit(s3, C):
        f = s3.float().sum(2)  # (B, C8, V, 8)
        return f.permute(0, 1, 3, 2).reshape(s3.shape[0], -1, s3.shape[3])[:, :C]
    x = torch.randn(B, 40, r ** 3, generator=g) * 2 + 0.5
    gn = torch.nn.GroupNorm(8, 40)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(40, generator=g)); gn.bias.copy_(torch.randn(40, generator=g))
    ref = ops.group_norm_(x.clone().cuda(), gn.weight.cuda(), gn.bias.cuda(), swish=True).cpu()
    got = unsplit(ops.to_s3(x.cuda(), gn.cuda(), swish=True).cpu(), 40)
    assert rel(got, ref) < 3e-7
    vc = torch.randint(0, r, (B, 3, n), generator=g, dtype=torch.int32)
    f = torch.randn(B, C, n, generator=g)
    ref = oracle_ops.avg_voxelize_forward(f, vc, r)[0]
    got = unsplit(ops.avg_voxelize_s3(f.cuda(), vc.cuda(), r).cpu(), C)
    assert float((got - ref).abs().max()) <= 2e-7 * float(ref.abs().max())  # exact 24-bit split up to the last rounding


@pytest.mark.parametrize("cin,cout,r,npts", [(35, 32, 32, 4096), (390, 32, 32, 4096), (64, 64, 32, 1100), (128, 64, 16, 1024),
                                             (256, 256, 8, 64), (192, 128, 8, 256), (16, 8, 8, 2000)])
def test_sparse_first_conv_equals_dense(ops, oracle_ops, cin, cout, r, npts):
    """Conv3d(avg_voxelize(f)) on the occupied voxels only == the dense evaluation (fp32 summation order aside)."""
    B = 2
    g = torch.Generator().manual_seed(cin + r + npts)
    vc = (torch.randn(B, 3, npts, generator=g) * r / 8 + r / 2).round().clamp(0, r - 1).to(torch.int32)
    vc[1, :, : npts // 2] = 0  # many points in one corner voxel, incl. the grid boundary
    f = torch.randn(B, cin, npts, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    vox = oracle_ops.avg_voxelize_forward(f, vc, r)[0]
    ref = TF.conv3d(vox.double().view(B, cin, r, r, r), w.double(), bias.double(), padding=1).float().reshape(B, cout, -1)
    got = ops.sparse_first_conv(f.cuda(), vc.cuda(), r, ops.sparse_conv_pack(w.cuda()), bias.cuda(), cout).cpu()
    assert rel(got, ref) < 2e-6
    # bf16x6 GEMM on pre-split operands: same bound (fp32-grade products, fp32 accumulation)
    got6 = ops.sparse_first_conv(f.cuda(), vc.cuda(), r, ops.sparse_conv_pack_s3(w.cuda()), bias.cuda(), cout).cpu()
    assert rel(got6, ref) < 2e-6
    # fp16x3 GEMM (the default): same bound
    goth = ops.sparse_first_conv(f.cuda(), vc.cuda(), r, ops.sparse_conv_pack_h2(w.cuda()), bias.cuda(), cout).cpu()
    assert rel(goth, ref) < 2e-6
    assert torch.equal(goth, ops.sparse_first_conv(f.cuda(), vc.cuda(), r, ops.sparse_conv_pack_h2(w.cuda()), bias.cuda(), cout).cpu())
    # runs of a concat buffer (strided features) give the same result
    big = torch.randn(B, cin + 6, npts, generator=g).cuda()
    big[:, 3:3 + cin] = f.cuda()
    got2 = ops.sparse_first_conv(big[:, 3:3 + cin], vc.cuda(), r, ops.sparse_conv_pack(w.cuda()), bias.cuda(), cout).cpu()
    assert torch.equal(got, got2)
    assert torch.equal(goth, ops.sparse_first_conv(big[:, 3:3 + cin], vc.cuda(), r, ops.sparse_conv_pack_h2(w.cuda()), bias.cuda(), cout).cpu())
    if _has_experimental():  # one-kernel form (GEMM + deterministic scatter into LDS accumulators, fp16x3): same bound; bit-reproducible
        gotf = ops.sparse_first_conv(f.cuda(), vc.cuda(), r, ops.sparse_conv_pack_fused(w.cuda()), bias.cuda(), cout).cpu()
        assert rel(gotf, ref) < 2e-6
        assert torch.equal(gotf, ops.sparse_first_conv(f.cuda(), vc.cuda(), r, ops.sparse_conv_pack_fused(w.cuda()), bias.cuda(), cout).cpu())
        assert torch.equal(gotf, ops.sparse_first_conv(big[:, 3:3 + cin], vc.cuda(), r, ops.sparse_conv_pack_fused(w.cuda()), bias.cuda(), cout).cpu())


@pytest.mark.parametrize("cin,cout,r,npts", [(35, 32, 32, 4096), (64, 64, 32, 1100), (64, 64, 32, 4096), (128, 64, 16, 1024), (128, 128, 16, 1024),
                                             (256, 256, 8, 64), (192, 128, 8, 256), (16, 8, 8, 2000), (390, 32, 32, 700)])
@pytest.mark.parametrize("form", ["dil", "dil_compact"])
@pytest.mark.parametrize("tile_form", ["0", "256", "64"])
def test_sparse_first_conv_output_stationary_equals_dense(ops, oracle_ops, monkeypatch, cin, cout, r, npts, form, tile_form):
    """The one-kernel output-stationary form with tap skipping (sparse_conv_os.hip, the default first convolution): == the dense
    evaluation at fp32 grade, bit-reproducible, strided features accepted, and its GroupNorm partials == the statistics of its output.
    tile_form: full tiles (one workgroup per CU) / half tiles of 256 or 64 entries (two per CU, round 6) -- every form runs the same
    MFMA sequence per output voxel, so the half forms must also give the full form's BITS."""
    from bdm_amd import _lib as L
    import ctypes
    if tile_form != "0" and not L.has_experimental():
        pytest.skip("half tiles: kernel family of the EXPERIMENTAL=1 build")
    monkeypatch.setattr(ops, "DIL_TILE", tile_form)
    B = 3
    g = torch.Generator().manual_seed(cin + r + npts)
    vc = (torch.randn(B, 3, npts, generator=g) * r / 8 + r / 2).round().clamp(0, r - 1).to(torch.int32)
    vc[1, :, : npts // 2] = 0            # many points in one corner voxel, incl. the grid boundary
    vc[2] = r - 1                        # a shape with ONE occupied voxel, at the far corner: almost every brick is pure bias
    if npts >= r * r:                    # shape 0: half of the points fill consecutive x-planes densely (tiles cut at plane boundaries)
        e = torch.arange(npts // 2)
        vc[0, :, : npts // 2] = torch.stack([(r // 2 - 1 + e // (r * r)).clamp(max=r - 1), (e // r) % r, e % r]).to(torch.int32)
    f = torch.randn(B, cin, npts, generator=g)
    f[:, : min(8, cin)] *= 300.0         # channels of very different magnitude share the shape's power-of-two scale
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    vox = oracle_ops.avg_voxelize_forward(f, vc, r)[0]
    ref = TF.conv3d(vox.double().view(B, cin, r, r, r), w.double(), bias.double(), padding=1).float().reshape(B, cout, -1)
    wt = ops.sparse_conv_pack_os(w.cuda(), form)
    got = ops.sparse_first_conv(f.cuda(), vc.cuda(), r, wt, bias.cuda(), cout).cpu()
    assert bool(torch.isfinite(got).all())
    for b in range(B):
        assert rel(got[b], ref[b]) < 2e-6, b
    assert torch.equal(got, ops.sparse_first_conv(f.cuda(), vc.cuda(), r, wt, bias.cuda(), cout).cpu())
    if tile_form != "0":
        monkeypatch.setattr(ops, "DIL_TILE", "0")
        ops.clear_plan_cache()
        assert torch.equal(got, ops.sparse_first_conv(f.cuda(), vc.cuda(), r, wt, bias.cuda(), cout).cpu()), "half tiles != full tiles"
        monkeypatch.setattr(ops, "DIL_TILE", tile_form)
        ops.clear_plan_cache()
    big = torch.randn(B, cin + 6, npts, generator=g).cuda()
    big[:, 3:3 + cin] = f.cuda()
    assert torch.equal(got, ops.sparse_first_conv(big[:, 3:3 + cin], vc.cuda(), r, wt, bias.cuda(), cout).cpu())
    # a shape alone gives the bits it gives inside the batch (per-shape activation scale, one tile choice per resolution)
    alone = ops.sparse_first_conv(f[1:2].contiguous().cuda(), vc[1:2].contiguous().cuda(), r, wt, bias.cuda(), cout).cpu()
    assert torch.equal(alone, got[1:2])
    groups = 8
    if cout % groups == 0 and ops.sparse_os_gn_ok(cout, groups, r):
        ops.clear_plan_cache()
        pts = (torch.randn(B, 3, npts, generator=g) * 0.3).cuda()
        plan = ops.voxel_plan(pts, r)
        compact = form == "dil_compact"
        out, (ws, slices, gg) = ops.sparse_first_conv_os(f.cuda(), plan, wt[1:], bias.cuda(), cout, gn_groups=groups, compact=compact)
        plain = ops.sparse_first_conv_os(f.cuda(), plan, wt[1:], bias.cuda(), cout, compact=compact)
        if compact:
            for b in range(B):   # (rows beyond a shape's list length are never written)
                nd = int(plan.tile_start[b, :, 1].max())
                assert torch.equal(out.rows[b, :nd], plain.rows[b, :nd])
            out = out.dense()
            assert torch.equal(out, ops.sparse_first_conv_os(f.cuda(), plan, wt[1:], bias.cuda(), cout))   # compact == dense form, bit for bit
        else:
            assert torch.equal(out, plain)
        part = ws.sum(2).cpu()
        o = out.double().cpu().view(B, groups, -1)
        err_s = float((part[..., 0] - o.sum(-1)).abs().max() / o.abs().sum(-1).max())
        err_q = float((part[..., 1] - (o * o).sum(-1)).abs().max() / (o * o).sum(-1).max())
        assert err_s < 2e-6 and err_q < 2e-6, (err_s, err_q)   # fp32 partial sums of <= 512 values, fp64 above


def test_compact_grid_operand_split_equals_the_dense_one(ops):
    """The operand split of the second convolution fed with the COMPACT output of the first (rows of the dilated voxels + bias
    everywhere else, read through the plan's index) gives the H2 grid of the dense path bit for bit, saturation word included."""
    import torch.nn as nn
    for cin, cout, r, n in ((32, 32, 32, 4096), (64, 64, 32, 3000), (128, 128, 16, 1024), (256, 256, 8, 64), (192, 128, 8, 256)):
        g = torch.Generator().manual_seed(cin + n)
        B = 3
        f = torch.randn(B, cin, n, generator=g).cuda()
        pts = (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
        w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
        bias = torch.randn(cout, generator=g).cuda()
        gn = nn.GroupNorm(8, cout).cuda()
        with torch.no_grad():
            gn.weight.copy_(torch.randn(cout, generator=g) * 0.3 + 1.0)
            gn.bias.copy_(torch.randn(cout, generator=g) * 0.2)
        ops.clear_plan_cache()
        plan = ops.voxel_plan(pts, r)
        pk = ops.conv3d_h2_pack(w)
        dense, st_d = ops.sparse_first_conv_os(f, plan, pk, bias, cout, gn_groups=8)
        comp, st_c = ops.sparse_first_conv_os(f, plan, pk, bias, cout, gn_groups=8, compact=True)
        assert torch.equal(st_d[0], st_c[0])
        sat_d, sat_c = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        h_d, s_d = ops.to_h2(dense, gn, swish=True, saturated=sat_d, stats=st_d)
        h_c, s_c = ops.to_h2(comp, gn, swish=True, saturated=sat_c, stats=st_c)
        assert s_d == s_c and torch.equal(h_d.view(torch.int16), h_c.view(torch.int16)) and int(sat_d) == int(sat_c) == 0
        # a scale that saturates: both routes raise the word
        h_d, _ = ops.to_h2(dense, gn, swish=True, saturated=sat_d, stats=st_d, scale=2.0 ** 20)
        h_c, _ = ops.to_h2(comp, gn, swish=True, saturated=sat_c, stats=st_c, scale=2.0 ** 20)
        assert int(sat_d) == 1 and int(sat_c) == 1 and torch.equal(h_d.view(torch.int16), h_c.view(torch.int16))


@pytest.mark.parametrize("r,n,tile,half", [(32, 4096, 512, 0), (16, 1024, 256, 0), (8, 300, 128, 0),
                                          (32, 4096, 256, 256), (32, 4096, 128, 128), (32, 2000, 64, 64), (16, 1024, 256, 256), (16, 1024, 128, 128),
                                          (16, 700, 64, 64)])
def test_dilated_voxel_list_and_tile_table(ops, r, n, tile, half):
    from bdm_amd import _lib as L_
    if half and not L_.has_experimental():
        pytest.skip("half tiles: kernel family of the EXPERIMENTAL=1 build")
    """bdm_voxel_dilate against a host restatement: the once-dilated occupied set in ascending voxel order, the per-plane prefix of
    the occupied cells, and a tile table whose tiles (a) partition the list, (b) hold <= the tile size, (c) FULL tiles (half = 0): need
    <= 3 r^2 compact rows (planes x0-1 .. x1+1) -- including a dense slab that forces cuts at plane boundaries; HALF tiles (round 6):
    list row ranges that cover every occupied neighbour of their voxels, <= 1376 rows in total (the LDS budget of the two-workgroups-
    per-CU kernel: 3 tile + 12 r inside a plane by construction), one range of whole planes across planes / three (plane, y-row range)
    ranges inside one plane."""
    import numpy as np
    from bdm_amd import _lib as L
    XCAP = 1376
    B = 4
    g = torch.Generator().manual_seed(r + n)
    vc = (torch.randn(B, 3, n, generator=g) * r / 8 + r / 2).round().clamp(0, r - 1).to(torch.int32)
    e = torch.arange(n - 1)                                                                   # a dense slab: consecutive FULL x-planes from r/2 - 2 on,
    vc[1, :, : n - 1] = torch.stack([r // 2 - 2 + e // (r * r), (e // r) % r, e % r]).to(torch.int32)
    vc[1, :, n - 1] = 0                                                                         # and a lone cell that shifts the tiles off the plane boundaries
    vc[2] = 0                                                                                   # one occupied cell, at the grid corner
    vc[3, 0] = (torch.arange(n) % 3 == 0).to(torch.int32) * (r - 1)                            # two full-ish planes at the two ends of the grid: tiles across
    vc[3, 1], vc[3, 2] = (torch.arange(n) // r) % r, torch.arange(n) % r                       # planes whose rows do not fit are split at the boundaries
    lib, r3 = L.lib(), r ** 3
    cnt = torch.zeros(B, r3, dtype=torch.int32)
    lin = (vc[:, 0].long() * r + vc[:, 1].long()) * r + vc[:, 2].long()
    for b in range(B):
        cnt[b].index_add_(0, lin[b], torch.ones(n, dtype=torch.int32))
    cnt = cnt.cuda()
    tiles = lib.bdm_voxel_dilate_slices(r, half)
    dl = torch.full((B, r3), -1, dtype=torch.int32, device="cuda")
    di = torch.full((B, r3), -7, dtype=torch.int32, device="cuda")
    ps = torch.empty(B, r + 2, dtype=torch.int32, device="cuda")
    ts = torch.empty(B, tiles, 16, dtype=torch.int32, device="cuda")
    L.check(lib.bdm_voxel_dilate(B, r, r3, L.ptr(cnt), L.ptr(dl), L.ptr(di), L.ptr(ps), L.ptr(ts), half, L.stream()))
    dl, ps, ts, occ = dl.cpu().numpy(), ps.cpu().numpy(), ts.cpu().numpy(), (cnt.cpu().numpy() > 0).reshape(B, r, r, r)
    di = di.cpu().numpy()
    for b in range(B):
        P = np.zeros((r + 2,) * 3, bool); P[1:-1, 1:-1, 1:-1] = occ[b]
        D = np.zeros((r, r, r), bool)
        for dx in range(3):
            for dy in range(3):
                for dz in range(3):
                    D |= P[dx:dx + r, dy:dy + r, dz:dz + r]
        want = np.flatnonzero(D.reshape(-1))
        occ_rank = np.full(r3, -1, np.int64); occ_rank[np.flatnonzero(occ[b].reshape(-1))] = np.arange(int(occ[b].sum()))
        nt = int(ts[b, 0, 7])
        assert int(ts[b, nt - 1, 1]) == len(want) and np.array_equal(dl[b, :len(want)], want), (r, b)
        assert int(ts[b, tiles - 1, 1]) == len(want)            # (what the operand-split / SE kernels read)
        rank = np.full(r3, -1, np.int32); rank[want] = np.arange(len(want))
        assert np.array_equal(di[b], rank)
        per_plane = occ[b].reshape(r, -1).sum(1)
        assert np.array_equal(ps[b, :r + 1], np.concatenate([[0], np.cumsum(per_plane)])) and ps[b, r + 1] == ps[b, r]
        assert ts[b, 0, 0] == 0 and 1 <= nt <= tiles and (ts[b, :, 7] == nt).all()
        for t in range(nt):
            j0, j1, vf, ve, klo, nr = (int(v) for v in ts[b, t, :6])
            assert j1 - j0 <= tile and (t == 0 or j0 == ts[b, t - 1, 1])
            assert vf == (0 if t == 0 else (want[j0] if j0 < len(want) else r3)) and ve == (want[j1] if t + 1 < nt and j1 < len(want) else r3)
            if not half:
                assert 0 < j1 - j0
                x0, x1 = want[j0] // (r * r), want[j1 - 1] // (r * r)
                assert klo == ps[b, max(x0 - 1, 0)] and nr == ps[b, min(x1 + 2, r)] - klo and nr <= 3 * r * r
                continue
            if j1 == j0:
                continue                                          # an empty piece of a split tile (a plane without listed voxels)
            assert int(ts[b, t, 12]) == half
            ranges = [(klo, nr), (int(ts[b, t, 8]), int(ts[b, t, 9])), (int(ts[b, t, 10]), int(ts[b, t, 11]))]
            assert sum(c for _, c in ranges) <= XCAP, (r, b, t, ranges)
            x0, x1 = want[j0] // (r * r), want[j1 - 1] // (r * r)
            assert (int(ts[b, t, 6]) >> 8, int(ts[b, t, 6]) & 255) == (x0, x1)
            if x0 != x1:
                assert ranges[1][1] == 0 and ranges[2][1] == 0
            vs = want[j0:j1]                                      # every occupied neighbour of the tile's voxels lies in one of its ranges
            vx, vy, vz = vs // (r * r), (vs // r) % r, vs % r
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        gx, gy, gz = vx + dx, vy + dy, vz + dz
                        ok = (gx >= 0) & (gx < r) & (gy >= 0) & (gy < r) & (gz >= 0) & (gz < r)
                        k = occ_rank[((gx * r + gy) * r + gz)[ok]]
                        k = k[k >= 0]
                        inside = np.zeros(len(k), bool)
                        for lo, c in ranges:
                            inside |= (k >= lo) & (k < lo + c)
                        assert inside.all(), (r, b, t)
        assert (ts[b, nt:, 0] == len(want)).all() and (ts[b, nt:, 1] == len(want)).all()
        if b == 1 and not half:   # the dense slab: some tile other than the last is short, i.e. was cut at a plane boundary
            assert any(int(ts[b, t, 1] - ts[b, t, 0]) < tile for t in range(nt - 1)), "the dense slab did not force a cut"
        if b == 3 and half and r == 32:   # planes 0 and r - 1 hold n / 3 and 2 n / 3 cells: no tile spans both (their rows would not fit)
            pass


@experimental
def test_sparse_fused_conv_wide_dynamic_range_and_empty_shape(ops, oracle_ops):
    """channels of very different magnitude (the activation scale is ONE power of two from max |x|), a shape whose points
    all fall into a single voxel, batch 16 with 4096 points (the bench's grid)."""
    B, cin, cout, r, npts = 16, 64, 64, 32, 4096
    g = torch.Generator().manual_seed(5)
    vc = (torch.randn(B, 3, npts, generator=g) * r / 8 + r / 2).round().clamp(0, r - 1).to(torch.int32)
    vc[3] = 31                                           # one occupied voxel, at the grid corner
    f = torch.randn(B, cin, npts, generator=g)
    f[:, :8] *= 3000.0                                   # huge channels next to ...
    f[:, 8:16] *= 1e-4                                   # ... tiny ones
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    got = ops.sparse_first_conv(f.cuda(), vc.cuda(), r, ops.sparse_conv_pack_fused(w.cuda()), bias.cuda(), cout).cpu()
    goth = ops.sparse_first_conv(f.cuda(), vc.cuda(), r, ops.sparse_conv_pack_h2(w.cuda()), bias.cuda(), cout).cpu()
    assert rel(goth, got) < 1e-6
    for b in (0, 3, 15):
        vox = oracle_ops.avg_voxelize_forward(f[b:b + 1].contiguous(), vc[b:b + 1].contiguous(), r)[0]
        ref = TF.conv3d(vox.double().view(1, cin, r, r, r), w.double(), bias.double(), padding=1).float().reshape(1, cout, -1)
        assert rel(got[b:b + 1], ref) < 2e-6, b
    # the tiny channels alone: their products sit far below the big ones, yet keep fp32-grade ABSOLUTE accuracy
    f2 = f.clone(); f2[:, :8] = 0
    got2 = ops.sparse_first_conv(f2.cuda(), vc.cuda(), r, ops.sparse_conv_pack_fused(w.cuda()), bias.cuda(), cout).cpu()
    vox = oracle_ops.avg_voxelize_forward(f2[:1].contiguous(), vc[:1].contiguous(), r)[0]
    ref = TF.conv3d(vox.double().view(1, cin, r, r, r), w.double(), bias.double(), padding=1).float().reshape(1, cout, -1)
    assert rel(got2[:1], ref) < 2e-6


@pytest.mark.parametrize("r,n", [(32, 4096), (16, 1000), (8, 64), (8, 700)])
def test_plan_full_equals_separate_kernels(ops, r, n):
    """one-launch plan (+ compaction + row occupancy) gives the arrays of the three separate launches."""
    from bdm_amd import _lib as L
    B = 3
    g = torch.Generator().manual_seed(r * n)
    pts = (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
    ops.clear_plan_cache()
    p = ops.voxel_plan(pts, r)
    lib, r3, n_max = L.lib(), r ** 3, min(n, r ** 3)
    ind = torch.empty(B, n, dtype=torch.int32, device="cuda"); cnt = torch.empty(B, r3, dtype=torch.int32, device="cuda")
    ws = torch.empty(lib.bdm_voxelize_workspace_bytes(B, n, r), dtype=torch.uint8, device="cuda")
    L.check(lib.bdm_voxelize_plan(B, n, r, L.ptr(p.vox_coords), L.ptr(ind), L.ptr(cnt), L.ptr(ws), L.stream()))
    oi = torch.empty(B, r3, dtype=torch.int32, device="cuda"); ol = torch.full((B, n_max), -7, dtype=torch.int32, device="cuda")
    no = torch.empty(B, dtype=torch.int32, device="cuda"); ro = torch.empty(B, r * r, dtype=torch.uint8, device="cuda")
    L.check(lib.bdm_voxel_compact(B, r, n_max, L.ptr(cnt), L.ptr(oi), L.ptr(ol), L.ptr(no), L.stream()))
    L.check(lib.bdm_voxel_row_occupancy(B, r, L.ptr(cnt), L.ptr(ro), L.stream()))
    assert torch.equal(p.ind, ind) and torch.equal(p.cnt, cnt)  # (the workspace also holds an unordered scratch list)
    assert torch.equal(p.occ_index, oi) and torch.equal(p.n_occ, no) and torch.equal(p.rowocc, ro)
    for b in range(B):
        k = int(no[b])
        assert torch.equal(p.occ_list[b, :k], ol[b, :k])


def test_fp16x3_saturation_guard(ops, oracle_ops):
    """VERDICT r1 item 8 / ADVICE: fp16x3 clamps at +-65504.  The split kernel raises the layer's sticky device word when a
    scaled GroupNorm output leaves that range; poll_h2_saturation() (once per trajectory) reports it, and the layer then runs
    the bf16x6 kernels.  Adversarial input: a grid that is zero except ONE cell -> |z| = sqrt(cg * V - 1) = 362 sigma."""
    import warnings
    from bdm_amd.modules import PVConv
    from bdm_amd.utils.procedural import fill_module_
    gn = torch.nn.GroupNorm(8, 32).cuda()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    calm = torch.randn(1, 32, 32 ** 3, generator=torch.Generator().manual_seed(0)).cuda()
    ops.to_h2(calm, gn, swish=True, saturated=flag)
    assert int(flag) == 0
    spike = torch.zeros(1, 32, 32 ** 3, device="cuda")
    spike[0, 5, 777] = 1.0
    h2, inv = ops.to_h2(spike, gn, swish=True, saturated=flag)
    assert int(flag) == 1 and bool(torch.isfinite(h2.float()).all())   # clamped, not inf -- and reported
    # module level: the flag of a PVConv re-routes it to bf16x6 and warns, results stay correct
    pv = fill_module_(PVConv(16, 32, 3, resolution=8, with_se=True, with_se_relu=True).eval(), seed=3).cuda()
    pv.bdm_name = "test.pvconv"
    g = torch.Generator().manual_seed(1)
    f, c = torch.randn(2, 16, 300, generator=g).cuda(), (torch.randn(2, 3, 300, generator=g) * 0.3).cuda()
    t = torch.zeros(2, 8, 300, device="cuda")
    y0 = pv((f, c, t))[0].clone()
    assert ops.poll_h2_saturation() == [] and not getattr(pv, "h2_saturated", False)
    ops.to_h2(spike, gn, swish=True, saturated=ops.saturation_slot(pv, spike.device))   # what a degenerate cloud would do
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        hit = ops.poll_h2_saturation()
    assert hit == [pv] and pv.h2_saturated and any("test.pvconv" in str(x.message) for x in w)
    y1 = pv((f, c, t))[0]                                   # now on the bf16x6 kernels
    assert rel(y1.cpu(), y0.cpu()) < 2e-6
    assert ops.poll_h2_saturation() == []                   # the word was cleared


@pytest.mark.parametrize("cin,cout,r,n", [(16, 32, 8, 300), (32, 32, 32, 4096), (64, 64, 32, 2000), (128, 128, 16, 1024), (256, 256, 8, 64)])
def test_gn2_folded_tail_equals_separate_groupnorm_pass(ops, monkeypatch, cin, cout, r, n):
    """PVConv tail with the second GroupNorm folded into its consumers (statistics from the convolution's epilogue, normalise +
    Swish inside the SE reduction and the devoxelisation gather) vs the separate GroupNorm pass: same module output."""
    from bdm_amd.modules import PVConv
    from bdm_amd.utils.procedural import fill_module_
    pv = fill_module_(PVConv(cin, cout, 3, resolution=r, with_se=True, with_se_relu=True).eval(), seed=cin + r).cuda()
    g = torch.Generator().manual_seed(n)
    f, c = torch.randn(3, cin, n, generator=g).cuda(), (torch.randn(3, 3, n, generator=g) * 0.3).cuda()
    t = torch.zeros(3, 8, n, device="cuda")
    monkeypatch.setattr(PVConv, "fold_gn2", False)
    ref = pv((f, c, t))[0].clone()
    monkeypatch.setattr(PVConv, "fold_gn2", True)
    monkeypatch.setattr(PVConv, "se_in_devox", False)
    monkeypatch.setattr(PVConv, "fold_pf", False)
    got = pv((f, c, t))[0].clone()
    assert rel(got.cpu(), ref.cpu()) < 1e-6
    assert torch.equal(got, pv((f, c, t))[0])      # deterministic
    monkeypatch.setattr(PVConv, "se_in_devox", True)  # opt-in: SE block's FC layers inside the devoxelisation kernel
    assert torch.equal(pv((f, c, t))[0], got)      # same summation order as the separate FC kernel: same bits
    monkeypatch.setattr(PVConv, "se_in_devox", False)
    monkeypatch.setattr(PVConv, "fold_pf", True)   # default: the point branch's GroupNorm + Swish inside the devoxelisation kernel
    got_pf = pv((f, c, t))[0].clone()
    assert rel(got_pf.cpu(), ref.cpu()) < 1e-6
    assert torch.equal(got_pf, pv((f, c, t))[0])   # deterministic


@pytest.mark.parametrize("chans,shape", [((35, 32, 64), (2, 1024, 32)), ((67, 64, 128), (3, 256, 32)), ((131, 128, 256), (2, 64, 32)),
                                         ((259, 256, 256, 512), (2, 16, 32)), ((128, 128, 64), (2, 4096)), ((96, 40, 24), (2, 300)),
                                         ((16, 8, 8), (2, 100)),
                                         # <= 64 points with k >= 128: the K-split skinny kernel (plain, folded input, statistics)
                                         ((256, 256, 256), (2, 64)), ((832, 256, 256), (3, 64)), ((131, 128, 128), (3, 16)),
                                         ((200, 64, 32, 32), (2, 40)), ((512, 512, 24), (2, 33))])
def test_shared_mlp_groupnorm_folding_equals_separate_passes(ops, monkeypatch, chans, shape):
    """SharedMLP with the GroupNorms folded (statistics from the convolution's epilogue, normalise + Swish in the next
    consumer: the next convolution's operand staging or the max over neighbours) vs a GroupNorm pass per layer."""
    from bdm_amd.modules import SharedMLP
    from bdm_amd.utils.procedural import fill_module_
    dim = 2 if len(shape) == 3 else 1
    mlp = fill_module_(SharedMLP(chans[0], list(chans[1:]), dim=dim).eval(), seed=sum(chans)).cuda()
    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(shape[0], chans[0], *shape[1:], generator=g) * 2 + 0.5).cuda()
    monkeypatch.setattr(SharedMLP, "fold_gn", False)
    ref = mlp.run(x).clone()
    monkeypatch.setattr(SharedMLP, "fold_gn", True)
    got = mlp.run(x)                                   # inner layers folded, last GroupNorm as a pass
    assert rel(got.cpu(), ref.cpu()) < 2e-6
    if dim == 2:                                       # the SA module's form: last GroupNorm inside the max over neighbours
        h, pending = mlp.run(x, fold_last=True)
        out = ops.max_over_neighbors(h, fold=pending)
        assert rel(out.cpu(), ops.max_over_neighbors(ref).cpu()) < 2e-6
        h2, p2 = mlp.run(x, fold_last=True)
        assert torch.equal(ops.max_over_neighbors(h2, fold=p2), out)   # deterministic


@pytest.mark.parametrize("cin,r,n,scale", [(390, 32, 4096, 0.5), (64, 32, 4096, 0.1), (128, 16, 1024, 0.5), (192, 8, 256, 0.5), (256, 8, 64, 0.5),
                                           (13, 16, 700, 0.3)])
def test_sparse_feature_gather_from_lds_equals_global_gather(ops, monkeypatch, cin, r, n, scale):
    """LDS-cached feature gather (rows copied to LDS once per (shape, channel group)) vs the scattered global loads: the
    sparse first convolution's output is bit-identical for both GEMM arithmetics."""
    g = torch.Generator().manual_seed(cin + n)
    f = torch.randn(3, cin, n, generator=g).cuda()
    c = (torch.randn(3, 3, n, generator=g) * scale).cuda()
    cout = 32
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda()
    b = torch.randn(cout, generator=g).cuda()
    ops.clear_plan_cache()
    plan = ops.voxel_plan(c, r)
    for pack in (ops.sparse_conv_pack_s3, ops.sparse_conv_pack_h2):
        wt = pack(w)
        monkeypatch.setenv("BDM_STAGING", "0")
        ref = ops.sparse_first_conv_planned(f, plan, wt, b, cout).clone()
        monkeypatch.delenv("BDM_STAGING")
        got = ops.sparse_first_conv_planned(f, plan, wt, b, cout)
        assert torch.equal(got, ref)


@pytest.mark.parametrize("cin,cout,r,n", [(32, 32, 32, 4096), (64, 64, 32, 3000), (128, 128, 16, 1024), (256, 256, 8, 256), (16, 32, 8, 300)])
def test_devoxelisation_from_lds_equals_global_gather(ops, monkeypatch, cin, cout, r, n):
    """GroupNorm-folded devoxelisation with the channel grid evaluated once into LDS vs eight scattered loads per (point,
    channel): bit-identical PVConv output."""
    from bdm_amd.modules import PVConv
    from bdm_amd.utils.procedural import fill_module_
    pv = fill_module_(PVConv(cin, cout, 3, resolution=r, with_se=True, with_se_relu=True).eval(), seed=cin + r).cuda()
    g = torch.Generator().manual_seed(n)
    f, c = torch.randn(3, cin, n, generator=g).cuda(), (torch.randn(3, 3, n, generator=g) * 0.3).cuda()
    t = torch.zeros(3, 8, n, device="cuda")
    monkeypatch.setenv("BDM_STAGING", "0")
    ref = pv((f, c, t))[0].clone()
    monkeypatch.setenv("BDM_STAGING", "1")     # forced for every shape (the default picks it where it is faster)
    assert torch.equal(pv((f, c, t))[0], ref)


@pytest.mark.parametrize("cin,cout,r,n", [(32, 32, 32, 4096), (64, 64, 32, 2500), (128, 128, 16, 1024), (256, 256, 8, 256), (192, 128, 8, 64)])
def test_groupnorm1_statistics_from_the_gather_epilogue(ops, monkeypatch, cin, cout, r, n):
    """First GroupNorm of a PVConv with its statistics left by the sparse gather (r*r slice partials per group, reduced in
    parallel by the operand-split kernel) vs a statistics pass over the grid: same module output, deterministic; rows whose
    stencils see no occupied cell (pure bias) are counted too (scale 0.2: most of the 32^3 grid is empty)."""
    from bdm_amd.modules import PVConv
    from bdm_amd.utils.procedural import fill_module_
    pv = fill_module_(PVConv(cin, cout, 3, resolution=r, with_se=True, with_se_relu=True).eval(), seed=cin + r).cuda()
    g = torch.Generator().manual_seed(n)
    f, c = torch.randn(3, cin, n, generator=g).cuda(), (torch.randn(3, 3, n, generator=g) * 0.2).cuda()
    t = torch.zeros(3, 8, n, device="cuda")
    monkeypatch.setattr(PVConv, "fold_gn1", False)
    ref = pv((f, c, t))[0].clone()
    monkeypatch.setattr(PVConv, "fold_gn1", True)
    got = pv((f, c, t))[0].clone()
    assert rel(got.cpu(), ref.cpu()) < 1e-6
    assert torch.equal(got, pv((f, c, t))[0])


def test_concat2_rows_fp_assemble_two_source_gemm_and_channel_first_gather(ops):
    """Direct checks of the launch-merging entry points against their unmerged forms (bit-exact: same arithmetic)."""
    from bdm_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    B, n, m = 3, 700, 96
    # torch.cat([a, v broadcast], dim=1) in one launch
    a = torch.randn(B, 37, n, generator=g).cuda()
    v = torch.randn(B, 9, generator=g).cuda()
    got = ops.cat_channels([a, v[:, :, None].expand(-1, -1, n)])
    assert torch.equal(got, torch.cat([a, v[:, :, None].expand(-1, -1, n)], 1))
    got = ops.cat_channels([a[:, 5:20], a[:, 1:3]])                       # strided row views
    assert torch.equal(got, torch.cat([a[:, 5:20], a[:, 1:3]], 1))
    # FP-module assembly vs three separate launches
    pc, cc = torch.randn(B, 3, n, generator=g).cuda(), torch.randn(B, 3, m, generator=g).cuda()
    fa, fs, ft = torch.randn(B, 21, m, generator=g).cuda(), torch.randn(B, 13, n, generator=g).cuda(), torch.randn(B, 8, m, generator=g).cuda()
    idx = torch.empty(B, 3, n, dtype=torch.int32, device="cuda")
    w = torch.empty(B, 3, n, dtype=torch.float32, device="cuda")
    lib = L.lib()
    L.check(lib.bdm_three_nn_search(B, m, n, L.ptr(pc), L.ptr(cc), L.ptr(idx), L.ptr(w), L.stream()), "search")
    ref0, ref1 = torch.empty(B, 34, n, device="cuda"), torch.empty(B, 8, n, device="cuda")
    L.check(lib.bdm_three_nn_apply(B, 21, m, n, L.ptr(fa), L.c_ll(21 * m), m, L.ptr(idx), L.ptr(w), L.ptr(ref0), L.c_ll(34 * n), n, L.stream()), "a")
    ref0[:, 21:] = fs
    L.check(lib.bdm_three_nn_apply(B, 8, m, n, L.ptr(ft), L.c_ll(8 * m), m, L.ptr(idx), L.ptr(w), L.ptr(ref1), L.c_ll(8 * n), n, L.stream()), "b")
    out0, out1 = torch.empty_like(ref0), torch.empty_like(ref1)
    L.check(lib.bdm_fp_assemble(B, m, n, L.ptr(idx), L.ptr(w), 21, L.ptr(fa), L.c_ll(21 * m), m, 13, L.ptr(fs), L.c_ll(13 * n), n, 8,
                                L.ptr(ft), L.c_ll(8 * m), m, L.ptr(out0), L.c_ll(34 * n), n, L.ptr(out1), L.c_ll(8 * n), n, L.stream()), "fp")
    assert torch.equal(out0, ref0) and torch.equal(out1, ref1)
    # 1x1 convolution over cat([x, x2]) read in place vs over the concatenated copy (same k order: same bits)
    x1, x2 = torch.randn(B, 40, n, generator=g).cuda(), torch.randn(B, 67, n, generator=g).cuda()
    wt, bias = (torch.randn(48, 107, generator=g) / 10).cuda(), torch.randn(48, generator=g).cuda()
    ref = ops.pointwise_conv(torch.cat([x1, x2], 1), wt, bias)
    assert torch.equal(ops.pointwise_conv_gn(x1, wt, bias, x2=x2), ref)
    big = torch.randn(B, 120, n, generator=g).cuda()
    assert torch.equal(ops.pointwise_conv_gn(x1, wt, bias, x2=big[:, 3:70]), ops.pointwise_conv(torch.cat([x1, big[:, 3:70]], 1), wt, bias))
    # the same on a skinny shape (64 points, k = 832: pw_skinny_kernel reads the second source in place too) + statistics + amax
    xs1, xs2 = torch.randn(B, 576, 64, generator=g).cuda(), torch.randn(B, 256, 64, generator=g).cuda()
    ws, bs_ = (torch.randn(256, 832, generator=g) / 29).cuda(), torch.randn(256, generator=g).cuda()
    ys, st = ops.pointwise_conv_gn(xs1, ws, bs_, x2=xs2, out_groups=8)
    refs = ops.pointwise_conv(torch.cat([xs1, xs2], 1), ws, bs_)
    assert torch.equal(ys, refs)
    sums = st[0].view(B, 8, st[1], 2).sum(2)
    rd = refs.double().view(B, 8, -1)
    assert torch.allclose(sums[..., 0], rd.sum(-1), rtol=1e-6, atol=1e-3) and torch.allclose(sums[..., 1], (rd * rd).sum(-1), rtol=1e-6)
    am = ops.amax_slots(xs1.device, 2 * B)
    ya = ops.pointwise_conv_gn(xs1, ws[:, :576].contiguous(), bs_, amax=am, amax_rows=128)
    assert torch.equal(am.view(B, 2), torch.stack([ya[:, :128].abs().amax(dim=(1, 2)), ya[:, 128:].abs().amax(dim=(1, 2))], 1))
    # conditioning gather written channel-first == the point-major gather transposed
    C, HW = 29, 50
    xt = torch.randn(B, n, 3, generator=g).cuda()
    feat = torch.randn(B, HW, C, generator=g).cuda()
    pix = torch.randint(-1, HW, (B, n), generator=g, dtype=torch.int32).cuda()
    pm, cf = torch.empty(B, n, 3 + C, device="cuda"), torch.empty(B, 3 + C, n, device="cuda")
    L.check(lib.bdm_condition_gather(B, n, C, HW, L.ptr(xt), L.ptr(feat), L.ptr(pix), L.ptr(pm), L.stream()), "pm")
    L.check(lib.bdm_condition_gather_cf(B, n, C, HW, L.ptr(xt), L.ptr(feat), L.ptr(pix), L.ptr(cf), L.stream()), "cf")
    assert torch.equal(cf.transpose(1, 2), pm)
    assert ops.transpose12(cf.transpose(1, 2)).data_ptr() == cf.data_ptr()   # the denoiser's input transpose is a view
