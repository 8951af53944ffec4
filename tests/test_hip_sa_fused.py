"""The first set-abstraction module's grouped MLP by recomputation (csrc/sa_mlp_fused.hip, round 4) against the operator chain it
replaces (sa_group -> 1x1 GEMM + statistics -> folded 1x1 GEMM + statistics -> folded max) and against plain torch on the same
grouped tensor (pointnet.py:80-90, shared_mlp.py:11-37)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _module(c_in, m, n, B, seed):
    from bdm_amd.modules import PointNetSAModule
    torch.manual_seed(seed)
    sa = PointNetSAModule(m, 0.15, 32, in_channels=c_in, out_channels=[32, 64]).cuda()
    with torch.no_grad():
        for layer in sa.mlps[0].layers:
            if isinstance(layer, nn.GroupNorm):
                layer.weight.copy_(torch.randn_like(layer.weight) * 0.3 + 1.0)
                layer.bias.copy_(torch.randn_like(layer.bias) * 0.2)
    g = torch.Generator().manual_seed(seed + 1)
    coords = (torch.randn(B, 3, n, generator=g) * 0.3).cuda()
    features = torch.randn(B, c_in, n, generator=g).cuda()
    temb = torch.randn(B, 64, generator=g).cuda()[:, :, None].expand(-1, -1, n)
    return sa, features, coords, temb


@pytest.mark.parametrize("c_in,m,n,B", [(32, 1024, 2048, 3), (32, 200, 700, 2), (13, 64, 256, 1), (29, 9, 64, 2), (1, 40, 128, 2)])
def test_fused_grouped_mlp_equals_the_operator_chain(hip, c_in, m, n, B):
    from bdm_amd import ops
    sa, features, coords, temb = _module(c_in, m, n, B, seed=c_in + m)
    assert ops.sa_mlp2_fusable(sa.mlps[0], c_in, 32)
    with torch.no_grad():
        sa.fuse_mlp = True
        out_f, centers_f, temb_f = sa((features, coords, temb))
        again, _, _ = sa((features, coords, temb))
        sa.fuse_mlp = False
        out_c, centers_c, temb_c = sa((features, coords, temb))
        assert torch.equal(out_f, again)            # fixed-order statistics: bit-reproducible
        assert torch.equal(centers_f, centers_c) and torch.equal(temb_f, temb_c) and temb_f.stride(2) == 0
        assert out_f.shape == out_c.shape == (B, 64, m)
        for b in range(B):
            assert rel(out_f[b], out_c[b]) < 3e-6, b
        # plain torch on the grouped tensor
        idx = sa.query(coords, centers_f)
        grouped = ops.sa_group(coords, centers_f.contiguous(), features, idx)
        ref, layers = grouped, sa.mlps[0].layers
        for i in (0, 3):   # (the Swish modules of the HIP path refuse to run: the activation is fused everywhere)
            ref = torch.nn.functional.conv2d(ref, layers[i].weight, layers[i].bias)
            ref = torch.nn.functional.group_norm(ref, 8, layers[i + 1].weight, layers[i + 1].bias, layers[i + 1].eps)
            ref = ref * torch.sigmoid(ref)
        ref = ref.max(dim=-1).values
        for b in range(B):
            assert rel(out_f[b], ref[b]) < 1e-5, b


def test_fused_form_is_declined_for_other_widths(hip):
    from bdm_amd import ops
    from bdm_amd.modules import PointNetSAModule
    assert not ops.sa_mlp2_fusable(PointNetSAModule(256, 0.2, 32, in_channels=64, out_channels=[64, 128]).mlps[0], 64, 32)
    assert not ops.sa_mlp2_fusable(PointNetSAModule(256, 0.2, 16, in_channels=32, out_channels=[32, 64]).mlps[0], 32, 16)
    assert not ops.sa_mlp2_fusable(PointNetSAModule(16, 0.8, 32, in_channels=32, out_channels=[32, 64, 64]).mlps[0], 32, 32)
    assert not ops.sa_mlp2_fusable(PointNetSAModule(16, 0.8, 32, in_channels=61, out_channels=[32, 64]).mlps[0], 61, 32)


def test_a_shapes_result_does_not_depend_on_its_batch(hip):
    """The centres a wave walks follow the batch size (8 per wave at B = 16, 1 at B = 1); the GroupNorm statistics are summed along one
    binary tree over the centres whatever that number is: same bits alone and in a batch."""
    sa, features, coords, temb = _module(32, 1024, 2048, 16, seed=7)
    with torch.no_grad():
        sa.fuse_mlp = True
        batch, _, _ = sa((features, coords, temb))
        for row in (0, 5, 15):
            alone, _, _ = sa((features[row:row + 1].contiguous(), coords[row:row + 1].contiguous(), temb[row:row + 1]))
            assert torch.equal(alone[0], batch[row]), row
        four, _, _ = sa((features[4:8].contiguous(), coords[4:8].contiguous(), temb[4:8]))
        assert torch.equal(four, batch[4:8])
