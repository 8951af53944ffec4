"""Parity cases for the kernel variants that the BASELINE.json configurations actually dispatch:
  C2  B=16, N=4096  -> ball_query_kernel<4,8> (>= 16384 centres), batch-16 dense kernels, sparse conv at B=16;
  C3  Merging at N=4096, B=2 (mini schedule through PVCNN_fuse);
  C4  N=8192 with the 390-channel PC^2 denoiser;
  C5  N=16384: FPS with 16 points per thread and no LDS copy, PC^2 and PVD denoiser forwards.
The oracle (CPU) is run on the same seeded inputs; for the batch-16 forward on two sampled shapes (every operator is
per-shape: SURVEY.md 8e)."""
import pytest
import torch

from helpers import point_cloud_inputs, rel_l2, seeded

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_ball_query_c2_bench_shape_bit_exact(hip, oracle_ops):
    """B=16, N=4096, M=1024, r=0.1, U=32: exactly the SA0 query of the bench (ball_query_kernel<4,8>, 512 threads)."""
    B, n, m = 16, 4096, 1024
    pts = seeded((B, 3, n), 123, 0.3).contiguous()
    idx = oracle_ops.furthest_point_sampling(pts, m)
    ctr = oracle_ops.gather_features_forward(pts, idx)
    ref = oracle_ops.ball_query(ctr, pts, 0.1, 32)
    got = hip.ball_query(ctr.cuda(), pts.cuda(), 0.1, 32).cpu()
    assert torch.equal(ref, got)
    # and the FPS + gather that feed it, at batch 16
    assert torch.equal(idx, hip.furthest_point_sampling(pts.cuda(), m).cpu())


def test_ball_query_c5_shape_bit_exact(hip, oracle_ops):
    """C5's first level: N=16384, M=1024 at B=4 (4096 centres -> <4,4>) and B=16 rows of a ragged M (<4,8>)."""
    pts = seeded((4, 3, 16384), 321, 0.3).contiguous()
    idx = oracle_ops.furthest_point_sampling(pts, 1024)
    assert torch.equal(idx, hip.furthest_point_sampling(pts.cuda(), 1024).cpu())  # fps_kernel<16>, no LDS copy
    ctr = oracle_ops.gather_features_forward(pts, idx)
    assert torch.equal(oracle_ops.ball_query(ctr, pts, 0.1, 32), hip.ball_query(ctr.cuda(), pts.cuda(), 0.1, 32).cpu())
    pts16 = seeded((16, 3, 16384), 322, 0.3).contiguous()
    ctr16 = pts16[:, :, ::16][:, :, :1023].contiguous()
    assert torch.equal(oracle_ops.ball_query(ctr16, pts16, 0.1, 32), hip.ball_query(ctr16.cuda(), pts16.cuda(), 0.1, 32).cpu())


def test_pc2_forward_batch16_two_sampled_shapes(hip, oracle_ops):
    """One PC^2 forward at the bench's own size (B=16, N=4096); shapes 3 and 12 are checked against the oracle."""
    from bdm_amd.pvcnn import PVCNN2_PC2
    from bdm_amd.utils.procedural import fill_module_
    from oracle import ref_net
    net = fill_module_(PVCNN2_PC2(3, 64, extra_feature_channels=387).eval(), seed=13)
    B, N = 16, 4096
    x = point_cloud_inputs(B, 390, N, seed=515)
    t = torch.arange(B) * 61 + 7
    pick = [3, 12]
    ref = ref_net.pvcnn_forward(net.state_dict(), x[pick].contiguous(), t[pick])
    got = net.cuda()(x.cuda(), t.cuda()).cpu()
    assert rel_l2(got[pick], ref) < TOL
    for k, s in enumerate(pick):
        assert rel_l2(got[s], ref[k]) < TOL


def test_pvd_forward_batch16_two_sampled_shapes(hip, oracle_ops):
    from bdm_amd.pvcnn import PVCNN2_PVD
    from bdm_amd.utils.procedural import fill_module_
    from oracle import ref_net
    net = fill_module_(PVCNN2_PVD(3, 64, extra_feature_channels=0).eval(), seed=14)
    B, N = 16, 4096
    x = point_cloud_inputs(B, 3, N, seed=516)
    t = torch.arange(B) * 59 + 3
    pick = [0, 15]
    ref = ref_net.pvcnn_forward(net.state_dict(), x[pick].contiguous(), t[pick])
    got = net.cuda()(x.cuda(), t.cuda()).cpu()
    assert rel_l2(got[pick], ref) < TOL


@pytest.mark.parametrize("N,B", [(8192, 2), (16384, 1), (16384, 3)])
def test_pc2_large_point_counts(hip, oracle_ops, N, B):
    """C4 (N=8192) and C5 (N=16384) with the 390-channel PC^2 denoiser."""
    from bdm_amd.pvcnn import PVCNN2_PC2
    from bdm_amd.utils.procedural import fill_module_
    from oracle import ref_net
    net = fill_module_(PVCNN2_PC2(3, 64, extra_feature_channels=387).eval(), seed=15)
    x = point_cloud_inputs(B, 390, N, seed=600 + N + B)
    t = torch.tensor([17, 480, 960][:B])
    pick = [B - 1]
    ref = ref_net.pvcnn_forward(net.state_dict(), x[pick].contiguous(), t[pick])
    got = net.cuda()(x.cuda(), t.cuda()).cpu()
    assert rel_l2(got[pick], ref) < TOL


def test_pvd_c5_point_count(hip, oracle_ops):
    from bdm_amd.pvcnn import PVCNN2_PVD
    from bdm_amd.utils.procedural import fill_module_
    from oracle import ref_net
    net = fill_module_(PVCNN2_PVD(3, 64, extra_feature_channels=0).eval(), seed=16)
    x = point_cloud_inputs(2, 3, 16384, seed=777)
    t = torch.tensor([250, 3])
    ref = ref_net.pvcnn_forward(net.state_dict(), x, t)
    got = net.cuda()(x.cuda(), t.cuda()).cpu()
    assert rel_l2(got, ref) < TOL


def test_mini_merging_n4096_b2(hip, oracle_ops):
    """C3's shape (Merging, N=4096) on a short schedule, B=2: PC^2 steps, 1-step branches, one fused step, final step."""
    import trajectory_case as case
    c = case.build(4096, head_scale=1.0, milestones=[1000, 996, 993, 990], roll_step=2, merging=True, B=2, seed=21)
    ref = case.run_oracle(c)
    got = case.run_hip(c)
    assert rel_l2(got, ref) < 1e-3


def test_mini_blending_n8192_b2(hip, oracle_ops):
    """C4's shape (Blending, N=8192) on a short schedule, B=2."""
    import trajectory_case as case
    c = case.build(8192, head_scale=1.0, milestones=[1000, 997, 994, 992], roll_step=1, merging=False, B=2, seed=8)
    ref = case.run_oracle(c)
    got = case.run_hip(c)
    assert rel_l2(got, ref) < 1e-3
