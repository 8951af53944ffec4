"""Known-answer tests pinning the CPU oracle's restatement of the seven native ops
(hand-derived from experiments/model/pvcnn/modules/functional/src/**).
The reference has no tests for these ops, so these cases ARE the pin."""
import numpy as np
import torch


def T(a, dtype=torch.float32):
    return torch.tensor(a, dtype=dtype)


def test_ball_query_fill_strict_and_empty(oracle_ops):
    # points on the x axis at 0, 0.05, 0.1, 0.2, 1.0; radius 0.1 (r2 = float(0.1)*float(0.1))
    pts = T([[[0.0, 0.05, 0.1, 0.2, 1.0], [0] * 5, [0] * 5]])
    ctr = T([[[0.0, 5.0, 0.2], [0, 0, 0], [0, 0, 0]]])
    out = oracle_ops.ball_query(ctr, pts, 0.1, 4)
    # centre 0: hits k=0 (d2=0), k=1 (0.0025) ; k=2 has d2 = 0.1f*0.1f == r2 -> NOT < r2 (strict, ball_query.cu:39)
    assert out[0, 0].tolist() == [0, 1, 0, 0]  # padded with FIRST hit (ball_query.cu:40-44)
    assert out[0, 1].tolist() == [0, 0, 0, 0]  # no hit -> zeros (ball_query.cpp:20-22)
    # centre 2 at x=0.2: k=3 (d=0) and k=2: dx = 0.2f-0.1f = 0.1000000015 -> d2 > r2? compute in float
    dx = np.float32(0.2) - np.float32(0.1)
    hit2 = bool(np.float32(dx * dx) < np.float32(np.float32(0.1) * np.float32(0.1)))
    exp = [2, 3, 2, 2] if hit2 else [3, 3, 3, 3]
    assert out[0, 2].tolist() == exp


def test_ball_query_truncates_to_first_u_ascending(oracle_ops):
    n = 100
    pts = torch.zeros(1, 3, n)
    pts[0, 0] = torch.linspace(0, 0.01, n)
    ctr = torch.zeros(1, 3, 1)
    out = oracle_ops.ball_query(ctr, pts, 0.5, 8)
    assert out[0, 0].tolist() == list(range(8))


def test_fps_tie_rule_lane_then_index(oracle_ops):
    # all points at the origin except candidates at the same distance from point 0:
    # k=90 (lane 90) and k=600 (lane 88): lane 88 wins although 600 > 90 (sampling.cu:120-160)
    n = 1024
    c = torch.zeros(1, 3, n)
    c[0, 0, 90] = 1.0
    c[0, 0, 600] = -1.0
    idx = oracle_ops.furthest_point_sampling(c, 3)
    assert idx[0].tolist()[:2] == [0, 600]
    # then remaining farthest is k=90 (distance 1 from origin cluster, 4 from 600 -> min is 1)
    assert idx[0, 2].item() == 90
    # same lane (k mod 512 equal): the smaller k wins
    c = torch.zeros(1, 3, n)
    c[0, 1, 5] = 2.0
    c[0, 1, 517] = -2.0
    idx = oracle_ops.furthest_point_sampling(c, 2)
    assert idx[0].tolist() == [0, 5]


def test_fps_small_n_and_more_samples_than_points(oracle_ops):
    c = T([[[0.0, 1.0, 3.0], [0, 0, 0], [0, 0, 0]]])
    idx = oracle_ops.furthest_point_sampling(c, 5)
    # 0 -> farthest 2 -> then 1 (min dist 1) -> then all distances 0: argmax picks k=0
    assert idx[0].tolist() == [0, 2, 1, 0, 0]


def test_gather_and_grouping(oracle_ops):
    f = torch.arange(2 * 3 * 5, dtype=torch.float32).view(2, 3, 5)
    idx = T([[4, 0], [1, 1]], torch.int32)
    g = oracle_ops.gather_features_forward(f, idx)
    assert torch.equal(g, torch.stack([f[0][:, [4, 0]], f[1][:, [1, 1]]]))
    nb = T([[[0, 1], [4, 4]], [[2, 3], [0, 0]]], torch.int32)
    gg = oracle_ops.grouping_forward(f, nb)
    assert gg.shape == (2, 3, 2, 2)
    assert torch.equal(gg[1, :, 0, 1], f[1, :, 3])


def test_three_nn_weights_and_ties(oracle_ops):
    # centres at x = 0,1,2,3 ; point at x=0.5 : d = .25,.25,2.25,6.25 -> ties keep the EARLIER index first
    ctr = T([[[0.0, 1.0, 2.0, 3.0], [0] * 4, [0] * 4]])
    pts = T([[[0.5, 3.0], [0, 0], [0, 0]]])
    feat = T([[[10.0, 20.0, 30.0, 40.0]]])
    out, idx, w = oracle_ops.three_nearest_neighbors_interpolate_forward(pts, ctr, feat)
    assert idx[0, :, 0].tolist() == [0, 1, 2]
    d0, d1, d2 = np.float32(0.25), np.float32(0.25), np.float32(2.25)
    s = np.float32(1.0) / np.float32(np.float32(d0 * d1 + d0 * d2) + d1 * d2)
    exp_w = [np.float32(d1 * d2) * s, np.float32(d0 * d2) * s, np.float32(d0 * d1) * s]
    assert np.allclose(w[0, :, 0].numpy(), exp_w, rtol=0, atol=0)
    assert abs(float(w[0, :, 0].sum()) - 1.0) < 1e-6
    # point exactly on centre 3: distance 0 clamps to 1e-10 (neighbor_interpolate.cu:61-63) -> weight ~1
    assert idx[0, 0, 1].item() == 3 and float(w[0, 0, 1]) > 0.999999
    assert abs(float(out[0, 0, 1]) - 40.0) < 1e-4


def test_three_nn_point_on_two_coincident_centres_is_clamped(oracle_ops):
    """two centres AT the query point: both squared distances are 0 and clamp to 1e-10 (neighbor_interpolate.cu:61-63);
    without the clamp the weights would be 0/0."""
    ctr = T([[[1.0, 1.0, 4.0], [0] * 3, [0] * 3]])
    pts = T([[[1.0], [0], [0]]])
    feat = T([[[10.0, 30.0, 99.0]]])
    out, idx, w = oracle_ops.three_nearest_neighbors_interpolate_forward(pts, ctr, feat)
    assert idx[0, :, 0].tolist() == [0, 1, 2]
    d0 = d1 = np.float64(np.float32(1e-10))
    d2 = np.float64(9.0)
    p01, p02, p12 = np.float32(d0 * d1), np.float32(d0 * d2), np.float32(d1 * d2)
    inv = np.float32(1.0) / np.float32(np.float32(p01 + p02) + p12)
    assert w[0, :, 0].tolist() == [float(p12 * inv), float(p02 * inv), float(p01 * inv)]
    assert abs(float(out[0, 0, 0]) - 20.0) < 1e-4


def test_avg_voxelize_counts_mean_and_empty(oracle_ops):
    r = 2
    coords = T([[[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 1, 0]]], torch.int32)  # voxels 0,0,7,6
    feat = T([[[1.0, 3.0, 5.0, 7.0], [2.0, 2.0, 2.0, 2.0]]])
    out, ind, cnt = oracle_ops.avg_voxelize_forward(feat, coords, r)
    assert ind[0].tolist() == [0, 0, 7, 6]
    assert cnt[0].tolist() == [2, 0, 0, 0, 0, 0, 1, 1]
    assert out[0, 0].tolist() == [2.0, 0, 0, 0, 0, 0, 7.0, 5.0]
    assert out[0, 1].tolist() == [2.0, 0, 0, 0, 0, 0, 2.0, 2.0]


def test_avg_voxelize_order_is_ascending_point_index(oracle_ops):
    # three addends whose float sum depends on the order
    vals = np.array([1.0, 1e8, -1e8], dtype=np.float32)  # ascending: (1/3 + 1e8/3) - 1e8/3 = 0; descending: 1/3
    coords = torch.zeros(1, 3, 3, dtype=torch.int32)
    out, _, _ = oracle_ops.avg_voxelize_forward(T(vals).view(1, 1, 3), coords, 1)
    inv = np.float32(1.0 / 3.0)
    exp = np.float32(np.float32(np.float32(vals[0] * inv) + np.float32(vals[1] * inv)) + np.float32(vals[2] * inv))
    assert float(out[0, 0, 0]) == float(exp) == 0.0


def test_devoxelize_corner_rule_and_constant_grid(oracle_ops):
    r = 3
    grid = torch.arange(27, dtype=torch.float32).view(1, 1, 27)
    # integer coordinates: weight 1 on the lower corner, the +1 neighbours are NOT touched
    # (hi offset only if frac > 0, trilinear_devox.cu:64-75) -> safe at the upper boundary r-1
    coords = T([[[2.0, 0.5], [2.0, 0.0], [2.0, 1.25]]])
    out = oracle_ops.trilinear_devoxelize_forward(r, False, coords, grid)[0]
    assert float(out[0, 0, 0]) == 26.0
    # (0.5, 0, 1.25): x between planes 0,1 ; z between 1,2
    exp = 0.5 * (0.75 * 1 + 0.25 * 2) + 0.5 * (0.75 * 10 + 0.25 * 11)
    assert abs(float(out[0, 0, 1]) - exp) < 1e-6
    const = torch.full((1, 2, 27), 3.5)
    pts = torch.rand(1, 3, 50) * 2.0
    o = oracle_ops.trilinear_devoxelize_forward(r, False, pts.contiguous(), const)[0]
    assert torch.allclose(o, torch.full_like(o, 3.5), atol=1e-5)


def test_devoxelize_integer_coordinate_never_reads_the_upper_neighbour(oracle_ops):
    """frac == 0 on an axis: the '+1' cell is NOT addressed (trilinear_devox.cu:64-75) -- a NaN stored there must not
    reach the output (0 * NaN would)."""
    r = 3
    grid = torch.arange(27, dtype=torch.float32).view(1, 1, 27).clone()
    grid[0, 0, 1 * 9 + 1 * 3 + 2] = float("nan")   # cell (1,1,2) = z-neighbour of (1,1,1)
    grid[0, 0, 2 * 9 + 1 * 3 + 1] = float("nan")   # cell (2,1,1) = x-neighbour
    grid[0, 0, 1 * 9 + 2 * 3 + 1] = float("nan")   # cell (1,2,1) = y-neighbour
    coords = T([[[1.0], [1.0], [1.0]]])
    out = oracle_ops.trilinear_devoxelize_forward(r, False, coords, grid)[0]
    assert float(out[0, 0, 0]) == 13.0


def test_fps_equal_distances_in_one_lane_keep_the_first(oracle_ops):
    """stage 1 is a strict '>' scan (sampling.cu:128-131): of two equally far points handled by the same thread
    (k and k + 512) the smaller k wins; stage 2 keeps the LEFT lane on ties (:154)."""
    n = 1100
    c = torch.zeros(1, 3, n)
    c[0, 2, 7] = 3.0
    c[0, 2, 7 + 512] = -3.0
    c[0, 2, 8] = 3.0            # a neighbouring lane with the same distance: lane 7 (left) still wins
    assert oracle_ops.furthest_point_sampling(c, 2)[0].tolist() == [0, 7]


def test_properties_random(oracle_ops):
    g = torch.Generator().manual_seed(0)
    B, N, M = 2, 300, 40
    pts = torch.randn(B, 3, N, generator=g) * 0.3
    idx = oracle_ops.furthest_point_sampling(pts, M)
    ctr = oracle_ops.gather_features_forward(pts, idx)
    nb = oracle_ops.ball_query(ctr, pts, 0.2, 16).long()
    # ascending up to the padding, every listed index inside the radius
    r2 = np.float32(0.2) * np.float32(0.2)
    for b in range(B):
        for j in range(M):
            d2 = ((pts[b][:, nb[b, j]] - ctr[b][:, j:j + 1]) ** 2).sum(0)
            assert bool((d2 < r2 + 1e-6).all())
    # 3-NN weights sum to 1
    _, _, w = oracle_ops.three_nearest_neighbors_interpolate_forward(pts, ctr, torch.randn(B, 4, M, generator=g))
    assert torch.allclose(w.sum(1), torch.ones(B, N), atol=1e-5)
    # voxelize: sum_v out*cnt == sum_p feat
    r = 4
    vc = torch.randint(0, r, (B, 3, N), generator=g, dtype=torch.int32)
    f = torch.randn(B, 5, N, generator=g)
    out, ind, cnt = oracle_ops.avg_voxelize_forward(f, vc, r)
    assert torch.allclose((out * cnt[:, None].float()).sum(-1), f.sum(-1), atol=1e-3)
    assert int(cnt.sum()) == B * N
