"""GPU parity: the `_pvcnn_backend`-compatible HIP operators vs the CPU oracle, through the C ABI.
Bit-exact for every index output; float outputs are bit-exact too because the HIP kernels
use the oracle's unfused operation order (tolerance written per test)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cloud(B, N, seed, scale=0.4):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 3, N, generator=g) * scale).contiguous()


LEVELS = [(4096, 1024, 0.1), (1024, 256, 0.2), (256, 64, 0.4), (64, 16, 0.8)]


@pytest.mark.parametrize("n,m", [(4096, 1024), (1024, 256), (256, 64), (64, 16), (1100, 77), (8192, 1024),
                                 (16384, 64), (3, 5), (1, 1)])
def test_fps_bit_exact(hip, oracle_ops, n, m):
    B = 3
    pts = cloud(B, n, seed=n + m)
    if n >= 1024:  # force exact distance ties across lanes and within a lane (sampling.cu:120-160)
        pts[0, :, 90] = pts[0, :, 600] * 1.0
        pts[1, :, 5] = pts[1, :, 517]
    ref = oracle_ops.furthest_point_sampling(pts, m)
    got = hip.furthest_point_sampling(pts.cuda(), m).cpu()
    assert torch.equal(ref, got)


@pytest.mark.parametrize("n,m", [(4096, 300), (1024, 256), (700, 200), (130, 130), (2049, 64), (5000, 40), (12289, 24)])
def test_fps_lattice_ties_bit_exact(hip, oracle_ops, n, m):
    """Points on a coarse integer lattice: most rounds have MANY points at exactly the same distance (and duplicates: m exceeds
    the number of distinct points in the small cases, so the all-zero rounds are covered).  The winner must follow the
    reference's (k mod 512, k) preference whatever thread / wave / register slot of the sampler owns the point."""
    g = torch.Generator().manual_seed(n * 31 + m)
    pts = torch.randint(-2, 3, (2, 3, n), generator=g).float() * 0.25
    ref = oracle_ops.furthest_point_sampling(pts, m)
    got = hip.furthest_point_sampling(pts.cuda(), m).cpu()
    assert torch.equal(ref, got)


def test_fps_tie_rule_known_answer(hip):
    n = 1024
    c = torch.zeros(1, 3, n)
    c[0, 0, 90] = 1.0
    c[0, 0, 600] = -1.0
    assert hip.furthest_point_sampling(c.cuda(), 3).cpu()[0].tolist() == [0, 600, 90]


@pytest.mark.parametrize("n,m,radius", LEVELS + [(1100, 300, 0.15), (16384, 1024, 0.1), (50, 7, 1e-4)])
def test_ball_query_bit_exact(hip, oracle_ops, n, m, radius):
    B, U = 2, 32
    pts = cloud(B, n, seed=7 * n)
    idx = oracle_ops.furthest_point_sampling(pts, m)
    ctr = oracle_ops.gather_features_forward(pts, idx)
    if radius < 1e-3:
        ctr = ctr + 5.0  # no hits at all -> all-zero rows
    ref = oracle_ops.ball_query(ctr, pts, radius, U)
    got = hip.ball_query(ctr.cuda(), pts.cuda(), radius, U).cpu()
    assert torch.equal(ref, got)


@pytest.mark.parametrize("B,n,m,radius,scale", [(9, 4096, 1023, 0.1, 0.25), (16, 2000, 513, 0.1, 0.05), (5, 700, 600, 0.3, 1.0)])
def test_ball_query_multi_centre_waves_bit_exact(hip, oracle_ops, B, n, m, radius, scale):
    """large batches take the 4- and 2-centres-per-wave kernels; ragged m exercises the unused centre slots."""
    g = torch.Generator().manual_seed(n + m)
    pts = torch.randn(B, 3, n, generator=g) * scale
    ctr = pts[:, :, torch.randperm(n, generator=g)[:m]].contiguous()
    ref = oracle_ops.ball_query(ctr, pts, radius, 32)
    got = hip.ball_query(ctr.cuda(), pts.cuda(), radius, 32).cpu()
    assert torch.equal(ref, got)


def test_gather_grouping_exact(hip, oracle_ops):
    B, C, N, M, U = 2, 35, 1000, 130, 32
    g = torch.Generator().manual_seed(1)
    f = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, M), generator=g, dtype=torch.int32)
    nb = torch.randint(0, N, (B, M, U), generator=g, dtype=torch.int32)
    assert torch.equal(oracle_ops.gather_features_forward(f, idx), hip.gather_features_forward(f.cuda(), idx.cuda()).cpu())
    assert torch.equal(oracle_ops.grouping_forward(f, nb), hip.grouping_forward(f.cuda(), nb.cuda()).cpu())


@pytest.mark.parametrize("n,m,c", [(64, 16, 576), (256, 64, 320), (1024, 256, 320), (4096, 1024, 192), (1100, 3, 7),
                                   (500, 2, 4), (16384, 1024, 8)])
def test_three_nn_bit_exact(hip, oracle_ops, n, m, c):
    B = 2
    pts = cloud(B, n, seed=n)
    ctr = pts[:, :, torch.randperm(n, generator=torch.Generator().manual_seed(3))[:m]].contiguous()
    f = torch.randn(B, c, m, generator=torch.Generator().manual_seed(4))
    ro, ri, rw = oracle_ops.three_nearest_neighbors_interpolate_forward(pts, ctr, f)
    go, gi, gw = hip.three_nearest_neighbors_interpolate_forward(pts.cuda(), ctr.cuda(), f.cuda())
    assert torch.equal(ri, gi.cpu())
    assert torch.equal(rw, gw.cpu())  # tolerance: 0 (same unfused op order, IEEE division)
    assert torch.equal(ro, go.cpu())


@pytest.mark.parametrize("n,r,c", [(4096, 32, 35), (1024, 16, 128), (256, 8, 192), (64, 8, 256), (1100, 32, 3),
                                   (16384, 32, 4), (5000, 2, 3)])
def test_avg_voxelize_bit_exact(hip, oracle_ops, n, r, c):
    B = 2
    g = torch.Generator().manual_seed(n + r)
    # gaussian-concentrated voxel coordinates: many points per voxel near the centre
    vc = (torch.randn(B, 3, n, generator=g) * r / 8 + r / 2).round().clamp(0, r - 1).to(torch.int32)
    f = torch.randn(B, c, n, generator=g) * 100
    ro, ri, rc = oracle_ops.avg_voxelize_forward(f, vc, r)
    go, gi, gc = hip.avg_voxelize_forward(f.cuda(), vc.cuda(), r)
    assert torch.equal(ri, gi.cpu()) and torch.equal(rc, gc.cpu())
    assert torch.equal(ro, go.cpu())  # tolerance: 0 -- ascending-point-index accumulation on both sides
    # and run-to-run deterministic (the reference's float atomics are not)
    go2 = hip.avg_voxelize_forward(f.cuda(), vc.cuda(), r)[0]
    assert torch.equal(go, go2)


@pytest.mark.parametrize("n,r,c", [(4096, 32, 64), (1024, 16, 128), (64, 8, 256), (1100, 32, 5)])
def test_devoxelize_bit_exact(hip, oracle_ops, n, r, c):
    B = 2
    g = torch.Generator().manual_seed(n * r)
    coords = (torch.rand(B, 3, n, generator=g) * (r - 1)).contiguous()
    coords[:, :, :8] = coords[:, :, :8].round()  # integer coordinates incl. the upper face
    coords[0, :, 0] = r - 1
    grid = torch.randn(B, c, r ** 3, generator=g)
    ro = oracle_ops.trilinear_devoxelize_forward(r, False, coords, grid)[0]
    go = hip.trilinear_devoxelize_forward(r, False, coords.cuda(), grid.cuda())[0]
    assert torch.equal(ro, go.cpu())  # tolerance: 0


def test_argument_errors_raise_not_exit(hip):
    with pytest.raises(RuntimeError):
        hip.ball_query(torch.zeros(1, 3, 4), torch.zeros(1, 3, 8).cuda(), 0.1, 4)  # host tensor
    with pytest.raises(RuntimeError):
        hip.grouping_forward(torch.zeros(1, 3, 8).cuda(), torch.zeros(1, 2, 2).cuda())  # float indices
    from bdm_amd import _lib
    with pytest.raises(_lib.BdmHipError):
        hip.furthest_point_sampling(torch.zeros(1, 3, 20000).cuda(), 4)  # beyond the sampler's limit


# ---- every hand-derived known-answer case of tests/test_oracle_ops.py, run against the HIP backend ------------------
import test_oracle_ops as _KA  # noqa: E402

_KA_CASES = [n for n in dir(_KA) if n.startswith("test_")]


class _HipAsOps:
    """The `_pvcnn_backend` surface of the HIP library on CPU tensors (moves operands to the GPU and results back), so that
    the oracle's known-answer functions can be called with it unchanged."""

    def __init__(self, backend):
        self._b = backend

    def __getattr__(self, name):
        fn = getattr(self._b, name)

        def call(*args):
            dev = [a.cuda().contiguous() if torch.is_tensor(a) else a for a in args]
            out = fn(*dev)
            if isinstance(out, (list, tuple)):
                return [o.cpu() for o in out]
            return out.cpu()
        return call


@pytest.mark.parametrize("case", _KA_CASES)
def test_known_answer_cases_on_the_hip_backend(hip, case):
    """strict '<', first-hit fill, zero rows, FPS (k mod 512, k) ties, 3-NN ties / clamp, voxel summation order, the
    devoxelisation corner rule: the SAME known answers that pin the oracle, asked of the HIP kernels."""
    getattr(_KA, case)(_HipAsOps(hip))


def test_pvcnn_backend_extension_equals_oracle(hip, oracle_ops):
    """`import _pvcnn_backend` (the reference's module name): same calls the reference's functional/*.py make."""
    import importlib
    import os
    import sys
    from bdm_amd import _lib
    sys.path.insert(0, os.path.dirname(_lib.SO_PATH))
    try:
        B = importlib.import_module("_pvcnn_backend")
    finally:
        sys.path.pop(0)
    O = oracle_ops
    pts = cloud(2, 1100, seed=3)
    idx = B.furthest_point_sampling(pts.cuda(), 130)
    assert torch.equal(idx.cpu(), O.furthest_point_sampling(pts, 130))
    ctr = B.gather_features_forward(pts.cuda(), idx)
    nb = B.ball_query(ctr, pts.cuda(), 0.2, 32)
    assert torch.equal(nb.cpu(), O.ball_query(ctr.cpu(), pts, 0.2, 32))
    f = torch.randn(2, 7, 1100, generator=torch.Generator().manual_seed(1))
    assert torch.equal(B.grouping_forward(f.cuda(), nb).cpu(), O.grouping_forward(f, nb.cpu()))
    out, i3, w3 = B.three_nearest_neighbors_interpolate_forward(pts.cuda(), ctr, f[:, :, :130].contiguous().cuda())
    ro, ri, rw = O.three_nearest_neighbors_interpolate_forward(pts, ctr.cpu(), f[:, :, :130].contiguous())
    assert torch.equal(out.cpu(), ro) and torch.equal(i3.cpu(), ri) and torch.equal(w3.cpu(), rw)
    vc = torch.randint(0, 8, (2, 3, 1100), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    vo, vi, vcnt = B.avg_voxelize_forward(f.cuda(), vc.cuda(), 8)
    ro, ri, rc = O.avg_voxelize_forward(f, vc, 8)
    assert torch.equal(vo.cpu(), ro) and torch.equal(vi.cpu(), ri) and torch.equal(vcnt.cpu(), rc)
    nc = torch.rand(2, 3, 1100, generator=torch.Generator().manual_seed(4)) * 7
    d_out, d_i, d_w = B.trilinear_devoxelize_forward(8, True, nc.cuda(), vo)
    ro, ri, rw = O.trilinear_devoxelize_forward(8, True, nc, ro)
    assert torch.equal(d_out.cpu(), ro) and torch.equal(d_i.cpu(), ri) and torch.equal(d_w.cpu(), rw)
    gy = torch.randn(2, 7, 1100, generator=torch.Generator().manual_seed(5))
    assert torch.allclose(B.trilinear_devoxelize_backward(gy.cuda(), d_i, d_w, 8).cpu(), O.trilinear_devoxelize_backward(gy, ri, rw, 8), atol=1e-5)


@pytest.mark.parametrize("n,scale", [(4096, 0.5), (4096, 0.05), (8192, 0.4), (16384, 0.6), (1000, 1.5)])
def test_voxel_plan_eight_slab_kernel_equals_one_workgroup_kernel(monkeypatch, n, scale):
    """32^3 plan with eight workgroups per shape (each owns 4096 cells and derives the preceding points / occupied cells by
    itself) vs the one-workgroup-per-shape kernel: every output identical, also on a cloud squeezed into a few cells."""
    from bdm_amd import ops
    g = torch.Generator().manual_seed(n)
    coords = (torch.randn(3, 3, n, generator=g) * scale).cuda()
    fields = ("vox_coords", "ind", "cnt", "occ_index", "n_occ", "rowocc")
    plans = []
    for flag in ("0", "1"):
        monkeypatch.setenv("BDM_STAGING", flag)
        ops.clear_plan_cache()
        p = ops.voxel_plan(coords, 32)
        torch.cuda.synchronize()
        nb = p.ws.numel() // 4
        ws = p.ws.view(torch.int32)
        start, sorted_ = ws[:3 * 32768].clone(), ws[nb - 3 * n:].clone()     # VoxWs: start | tmp | sorted
        occ = [p.occ_list[b, :int(p.n_occ[b])].clone() for b in range(3)]
        plans.append(([getattr(p, f).clone() for f in fields], start, sorted_, occ))
    for a, b in zip(plans[0][0], plans[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(plans[0][1], plans[1][1]) and torch.equal(plans[0][2], plans[1][2])
    assert all(torch.equal(x, y) for x, y in zip(plans[0][3], plans[1][3]))
