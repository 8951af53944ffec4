"""Parity at the sizes the BASELINE.json configurations actually run per GPU.

(a) ONE full-length C2 trajectory -- BDM-Blending, real milestones, 1000 PC^2 + 80 PVD forwards + 5 blends -- on the HIP
    path at B = 16, N = 4096 in the per-shape Philox mode (`run.rng=per_shape`: exactly what bench.py times).  The CPU
    oracle re-runs ONE sampled shape of the batch from the same streams, restated on the host by oracle/ref_rng.py, and
    the final clouds are held to the literal north-star bound (relative L2 <= 1e-3).  HEAD_SCALE as in
    tests/test_hip_full_trajectory.py (chaos, not kernels, decides above it: DESIGN.md section 5).
(b) Per-GPU batches of C3 / C4 / C5: Merging on a mini schedule at B = 16, N = 4096; PC^2 forward at B = 8, N = 8192; PC^2
    and PVD forwards at B = 32, N = 16384.  Each: finite everywhere; one sampled shape against the oracle (<= 1e-4 per
    forward, <= 1e-3 on a final cloud); and that shape against ITSELF run alone at B = 1 on the HIP path -- batch
    invariance catches batch-dependent dispatch (grid.z = B, tile choices, `ball_query_kernel<4,8>` above 16384 centres,
    workspace sizing) without more oracle time.  Every operator of the path is per-shape and all
    power-of-two activation scales are per shape, so the only thing that may differ is the summation ORDER of the GroupNorm
    statistics where a convolution picks its tile from the number of workgroups the whole batch gives it (1-ulp effects on a
    mean / rstd): the bound is 1e-6 relative L2, two orders below the per-forward parity tolerance.  At EQUAL batch
    size a shape's bits do not depend on what its batch-mates contain (last test).
"""
import pytest
import torch

from helpers import first_segment_past, golden_trajectory, oracle_spread, parity, point_cloud_inputs, rel_l2
import trajectory_case as case

HEAD_SCALE = 0.1
NORTH_STAR = 1e-3
TOL_FORWARD = 1e-4
TOL_BATCH = 1e-6   # the same shape at B = 1 and inside its per-GPU batch
# Head scales of the two C2-size fixtures.  tests/test_hip_full_trajectory.py's rule -- the largest head scale of {1, 0.3, 0.1, 0.03, 0.01} at
# which a 1-ulp change of the initial cloud moves the final cloud by < 1e-4 -- was calibrated at N = 1024 (0.1: 1.9e-5 .. 4.9e-5).  At the
# bench's own size it gives 0.03: `tools/chaos_probe.py --hip --points 4096` measures 6.6e-4 at head scale 0.1 (a ONE-ulp perturbation ends
# as far away as the HIP path is from the oracle: 4.9e-4 .. 8.9e-4 depending on the summation order of the round's kernels) and 6.8e-5 at
# 0.03.  `tools/error_budget.py` (oracle/truth.py: the same network in float64 on fp32 geometry) splits the per-forward distance: HIP vs
# exact 0.8 - 1.1e-6, fp32 CPU oracle vs exact 0.75 - 0.83e-6 -- the HIP path is as close to exact arithmetic as the reference's own
# fp32 path, so at 0.1 the figure is the chaos of the procedural network times the fp32 noise floor of BOTH sides, not kernel error.
#   * head 0.03 (`c2_b16_shape11_h003`): THE bench-size parity test -- literal 1e-3 bound plus the early-warning line below;
#   * head 0.1  (`c2_b16_shape11`): a chaos monitor.  Its yardstick is INDEPENDENT of the product (round 6): the oracle's own
#     final cloud at a second reduction order (`traj_c2_b16_shape11_alt*.npz`: torch.set_num_threads(1) / (2) against the fixture's 8; one PC^2
#     forward differs 5 - 8e-7 between them) ends d_oo = 2.7e-4 from the fixture, against 1.4e-5 at head 0.03: the bound is
#     max(1e-3, 2 d_oo) = the literal 1e-3 today.  (Rounds 3 - 5: the HIP figure moved 4.9e-4 .. 8.9e-4 with the summation order of a round's
#     kernels; the same oracle on the GPU box's host gives 4.8e-4 where the build container's fixture gives 8.0e-4.)
C2_MARGIN_LINE = {"c2_b16_shape11_h003": 3e-4, "c2_b16_shape3_h003": 3e-4, "c2_b16_shape7_h003": 3e-4, "c2_b16_shape11": None}
_C2_RUNS = {}   # (head scale) -> (clouds, segment clouds) of the ONE B = 16 trajectory the fixtures of that head scale share


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2_b16_shape11_h003", "c2_b16_shape3_h003", "c2_b16_shape7_h003", "c2_b16_shape11"])
def test_full_c2_trajectory_batch16_vs_oracle_fixture(hip, name):
    """(a) in the default `-m gpu` selection: the oracle's cloud of the sampled shape is tests/golden/traj_<name>.npz
    (oracle/gen_golden_traj.py: the same case and streams, run once in the build container); the HIP side is the full B = 16,
    N = 4096, 1080-forward trajectory exactly as bench.py runs it."""
    g = golden_trajectory(name)
    B, N, seed, row = int(g["B"]), int(g["N"]), int(g["philox_seed"]), int(g["row"])
    assert (B, N) == (16, 4096)
    c = case.build(N, head_scale=float(g["head_scale"]), merging=False, B=B)
    assert list(g["milestones"]) == list(c.milestones) and len(case.program_order(c.milestones, c.roll_step)) == int(g["forwards"])
    key = (float(g["head_scale"]), seed)
    if key not in _C2_RUNS:     # three sampled shapes of the calm batch are checked against ONE run of it
        with case.segments() as seg:
            _C2_RUNS[key] = (case.run_hip_streams(c, seed, list(range(B))), seg.clouds)
    got, seg_clouds = _C2_RUNS[key]
    assert got.shape == (B, N, 3) and bool(torch.isfinite(got).all())
    line = C2_MARGIN_LINE[name]
    err = parity(f"traj_{name} final cloud (bench size, Philox mode, head {float(g['head_scale']):g})",
                 rel_l2(got[row:row + 1], torch.from_numpy(g["final"])), NORTH_STAR, note="" if line is None else f"margin line {line:.0e}")
    first, curve = first_segment_past(NORTH_STAR, [x[row:row + 1] for x in seg_clouds], g)
    for i, e in enumerate(curve):
        parity(f"traj_{name} segment {i}", e, NORTH_STAR)
    print(f"full C2 trajectory at B=16, N=4096 (per-shape Philox streams), head {float(g['head_scale']):g}, shape {row} vs the oracle fixture: "
          f"final rel-L2 {err:.3e}; per segment " + " ".join(f"{e:.1e}" for e in curve))
    bound = _monitor_bound(name, err) if line is None else NORTH_STAR
    assert err <= bound, (f"final rel-L2 {err:.3e} > {bound:.3e}; first schedule segment past 1e-3: {first} "
                          f"(segment curve {['%.2e' % e for e in curve]})")
    if line is not None:
        assert err <= line, (f"margin gone: final rel-L2 {err:.3e} is inside the 1e-3 bound but past the {line:.0e} early-warning line of the calm "
                             f"(head {float(g['head_scale']):g}) fixture; segment curve {['%.2e' % e for e in curve]}")
    assert rel_l2(got[(row + 1) % B:(row + 1) % B + 1], got[row:row + 1]) > 0.1


def _monitor_bound(name, err):
    """Bound of a head-0.1 chaos monitor: the literal 1e-3 or twice the ORACLE's own distance from itself at another reduction order
    (helpers.oracle_spread: fixtures only, no product code on either side), whichever is larger; the spread goes on the record."""
    d_oo = oracle_spread(name)
    assert d_oo is not None, f"tests/golden/traj_{name}_alt*.npz missing: python -m oracle.gen_golden_traj --threads 1 {name}"
    parity(f"traj_{name} oracle vs oracle at another reduction order (yardstick of the figure above)", d_oo, 1.0)
    bound = max(NORTH_STAR, 2.0 * d_oo)
    print(f"   oracle-vs-oracle spread {d_oo:.3e}; monitor bound {bound:.3e}; HIP / oracle spread {err / max(d_oo, 1e-30):.1f}x")
    return bound


# ---- C3's own per-GPU shape at FULL length (round 6) -----------------------------------------------------------------
C3_MARGIN_LINE = {"c3_b16_shape5_h003": 3e-4, "c3_b16_shape5": None}
_C3_RUNS = {}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c3_b16_shape5_h003", "c3_b16_shape5"])
def test_full_c3_merging_trajectory_batch16_vs_oracle_fixture(hip, name):
    """BDM-Merging at C3's per-GPU batch -- B = 16, N = 4096, per-shape Philox streams, the real milestones: 995 PC^2 + 75 PVD + 5 fused
    (PVCNN_fuse) forwards -- against the oracle's cloud of one sampled shape (tests/golden/traj_<name>.npz, oracle/gen_golden_traj.py;
    ~33 min of oracle time each, once).  Head 0.03 (the rule's scale at this size): literal 1e-3 + the 3e-4 early-warning line;
    head 0.1: chaos monitor against the oracle's own spread, as the C2-size one."""
    g = golden_trajectory(name)
    B, N, seed, row = int(g["B"]), int(g["N"]), int(g["philox_seed"]), int(g["row"])
    assert (B, N) == (16, 4096) and bool(g["merging"])
    c = case.build(N, head_scale=float(g["head_scale"]), merging=True, B=B)
    assert list(g["milestones"]) == list(c.milestones) and len(case.program_order(c.milestones, c.roll_step, True)) == int(g["forwards"]) == 1075
    with case.segments() as seg:
        got = case.run_hip_streams(c, seed, list(range(B)))
    assert got.shape == (B, N, 3) and bool(torch.isfinite(got).all())
    line = C3_MARGIN_LINE[name]
    err = parity(f"traj_{name} final cloud (C3 per-GPU batch, full length, Philox mode, head {float(g['head_scale']):g})",
                 rel_l2(got[row:row + 1], torch.from_numpy(g["final"])), NORTH_STAR, note="" if line is None else f"margin line {line:.0e}")
    first, curve = first_segment_past(NORTH_STAR, [x[row:row + 1] for x in seg.clouds], g)
    for i, e in enumerate(curve):
        parity(f"traj_{name} segment {i}", e, NORTH_STAR)
    print(f"full C3 (Merging) trajectory at B=16, N=4096, head {float(g['head_scale']):g}, shape {row} vs the oracle fixture: "
          f"final rel-L2 {err:.3e}; per segment " + " ".join(f"{e:.1e}" for e in curve))
    bound = _monitor_bound(name, err) if line is None else NORTH_STAR
    assert err <= bound, f"final rel-L2 {err:.3e} > {bound:.3e}; first schedule segment past 1e-3: {first} ({['%.2e' % e for e in curve]})"
    if line is not None:
        assert err <= line, f"margin gone: final rel-L2 {err:.3e} past the {line:.0e} early-warning line; segment curve {['%.2e' % e for e in curve]}"
    assert rel_l2(got[(row + 1) % B:(row + 1) % B + 1], got[row:row + 1]) > 0.1


@pytest.mark.gpu_slow
def test_full_c2_trajectory_batch16_sampled_shape_vs_oracle(hip, oracle_ops):
    B, N, seed, row = 16, 4096, 42, 11
    c = case.build(N, head_scale=HEAD_SCALE, merging=False, B=B)
    assert len(case.program_order(c.milestones, c.roll_step)) == 1080
    got = case.run_hip_streams(c, seed, list(range(B)))
    assert got.shape == (B, N, 3) and bool(torch.isfinite(got).all())
    ref = case.run_oracle(case.philox_shape_case(c, seed, row, row))
    err = parity(f"traj_live_c2_b16_shape{row} final cloud", rel_l2(got[row:row + 1], ref), NORTH_STAR)
    print(f"full C2 trajectory at B=16, N=4096 (per-shape Philox streams), shape {row} vs oracle: final rel-L2 {err:.3e}")
    assert err <= NORTH_STAR
    # the batch-mates are different samples (different streams, images, cameras): not one cloud repeated 16 times
    assert rel_l2(got[0:1], got[row:row + 1]) > 0.1


@pytest.mark.gpu
def test_c3_mini_merging_batch16(hip, oracle_ops):
    """C3's per-GPU batch (Merging, B = 16, N = 4096) on a short schedule through PVCNN_fuse."""
    B, seed, row = 16, 7, 13
    c = case.build(4096, head_scale=1.0, milestones=[1000, 996, 993, 990], roll_step=2, merging=True, B=B, seed=21)
    got = case.run_hip_streams(c, seed, list(range(B)))
    assert bool(torch.isfinite(got).all())
    ref = case.run_oracle(case.philox_shape_case(c, seed, row, row))
    err = parity("c3 mini-Merging B=16 N=4096: sampled shape vs oracle", rel_l2(got[row:row + 1], ref), NORTH_STAR)
    alone = case.run_hip_streams(case.subset(c, [row]), seed, [row])
    inv = parity("c3 mini-Merging B=16: shape vs itself at B=1", rel_l2(alone, got[row:row + 1]), 1e-4)
    print(f"mini BDM-Merging at B=16, N=4096: shape {row} vs oracle {err:.3e}; vs itself at B=1 {inv:.3e}")
    assert err <= NORTH_STAR
    assert inv <= 1e-4, f"batch-dependent result: rel-L2 {inv:.3e}"   # ten free-running steps at head scale 1 amplify the 1e-6 of a forward


def _forward_case(cls, B, N, extra, seed, row, monkeypatch):
    """-> (batch output, oracle output of shape `row`, that shape alone with the batch's kernel choice, alone with the default choice).
    The first convolution's FORM is picked per layer from (batch, points, resolution, channels) (ops.sparse_dil_pays: compact
    output-stationary kernel for a batch that fills the chip with tiles, GEMM + gather below that; likewise ops.compact_tail_pays for the
    rest of the voxel branch), so a shape run alone takes the
    other form on some layers -- same products, another fp32 summation order.  The batch-invariance property (no batch-dependent
    BUG: grid.z = B, workspace sizing, tile choices) is therefore tested with the form pinned to what the batch uses; the default
    choice at B = 1 is held to the forward tolerance class instead."""
    from bdm_amd.modules import PVConv
    from bdm_amd.utils.procedural import fill_module_
    from oracle import ref_net
    net = fill_module_(cls(3, 64, extra_feature_channels=extra).eval(), seed=seed)
    x = point_cloud_inputs(B, 3 + extra, N, seed=7000 + N + B)
    t = (torch.arange(B) * 31 + 5) % 1000
    ref = ref_net.pvcnn_forward(net.state_dict(), x[row:row + 1].contiguous(), t[row:row + 1])
    net = net.cuda()
    got = net(x.cuda(), t.cuda()).cpu()
    alone_default = net(x[row:row + 1].contiguous().cuda(), t[row:row + 1].cuda()).cpu()
    from bdm_amd import ops
    pays, tail = ops.sparse_dil_pays, ops.compact_tail_pays
    monkeypatch.setattr(ops, "sparse_dil_pays", lambda b, n, r, c: pays(B, n, r, c))   # the batch's choices, whatever the batch
    monkeypatch.setattr(ops, "compact_tail_pays", lambda b, n, r, c: tail(B, n, r, c))
    # (the voxel attention splits a query's keys into ranges when few shapes share a launch: bdm_attention_h2_key_slices -- the batch's count)
    from bdm_amd import _lib as L
    monkeypatch.setattr(ops, "ATTN_KSPLIT", int(L.lib().bdm_attention_h2_key_slices(B, 4096)))
    alone = net(x[row:row + 1].contiguous().cuda(), t[row:row + 1].cuda()).cpu()
    monkeypatch.undo()
    return got, ref, alone, alone_default


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,N,extra,row", [("c4_pc2", 8, 8192, 387, 5), ("c5_pc2", 32, 16384, 387, 29),
                                                ("c5_pvd", 32, 16384, 0, 17)])
def test_per_gpu_batch_forward(hip, oracle_ops, monkeypatch, name, B, N, extra, row):
    from bdm_amd.pvcnn import PVCNN2_PC2, PVCNN2_PVD
    got, ref, alone, alone_default = _forward_case(PVCNN2_PC2 if extra else PVCNN2_PVD, B, N, extra, 31 + B, row, monkeypatch)
    assert got.shape == (B, 3, N) and bool(torch.isfinite(got).all())
    err, inv, inv_d = rel_l2(got[row:row + 1], ref), rel_l2(alone, got[row:row + 1]), rel_l2(alone_default, got[row:row + 1])
    parity(f"forward {name} B={B} N={N}: sampled shape vs oracle", err, TOL_FORWARD)
    parity(f"forward {name}: shape vs itself at B=1 (same kernel forms)", inv, TOL_BATCH)
    parity(f"forward {name}: shape vs itself at B=1 (default forms)", inv_d, 10 * TOL_BATCH)
    print(f"{name}: B={B}, N={N}: shape {row} vs oracle {err:.3e}; vs itself at B=1 {inv:.3e} (same kernel forms), {inv_d:.3e} (default forms at B=1)")
    assert err < TOL_FORWARD
    assert inv <= TOL_BATCH, f"batch-dependent result: rel-L2 {inv:.3e}"
    assert inv_d <= 10 * TOL_BATCH and rel_l2(alone_default, ref) < TOL_FORWARD   # another summation order of the same fp16x3 products


@pytest.mark.gpu
def test_a_shape_does_not_see_its_batch_mates(hip):
    """the fp16x3 activation scales (sparse first convolution, attention q / k / v) are per SHAPE: at equal batch size a
    shape gives the same BITS whether its batch-mates are ordinary clouds or carry features 1000x larger / coordinates 100x smaller
    (C1-sized level: the 8^3 sparse GEMMs, the voxel attention and every other operator of both denoisers are on this path); and
    against the shape run alone (other tile choices: summation order of the GroupNorm statistics) it stays within 1e-6."""
    from bdm_amd.pvcnn import PVCNN2_PC2, PVCNN2_PVD
    from bdm_amd.utils.procedural import fill_module_
    for cls, extra in ((PVCNN2_PC2, 387), (PVCNN2_PVD, 0)):
        net = fill_module_(cls(3, 64, extra_feature_channels=extra).eval(), seed=3).cuda()
        x = point_cloud_inputs(3, 3 + extra, 1024, seed=91)
        t = torch.tensor([10, 500, 990])
        calm = net(x.cuda(), t.cuda()).cpu()
        wild = x.clone()
        wild[1] *= 1000.0
        wild[2, :3] *= 0.01
        got = net(wild.cuda(), t.cuda()).cpu()
        assert torch.equal(got[0], calm[0]), (cls.__name__, rel_l2(got[0], calm[0]))
        assert not torch.equal(got[1], calm[1])
        for s in range(3):
            alone = net(wild[s:s + 1].contiguous().cuda(), t[s:s + 1].cuda()).cpu()
            assert rel_l2(alone, got[s:s + 1]) <= TOL_BATCH, (cls.__name__, s, rel_l2(alone, got[s:s + 1]))


@pytest.mark.gpu
def test_recorded_step_equals_eager_loop_at_the_bench_batch(hip, monkeypatch):
    """The launch tape (default for every BASELINE configuration: model.TAPE_MAX_POINTS) replays the step out of a PRIVATE memory
    pool, i.e. with the block re-use of the eager loop but a host that runs ahead: a buffer allocated on one stream and read on
    another after its last Python reference is gone shows up here as a difference (the hoisted point-branch gather did, at
    B = 16 only).  Blending at B = 16, N = 4096 over two real segments (64 PC^2 steps, 16 PVD steps, one blend): bit-equal."""
    from bdm_amd import model as M
    from bdm_amd.config import ProjectConfig
    from bdm_amd.data import SyntheticShapes
    from bdm_amd.model import get_model
    from bdm_amd.pvd import prepare_pvd_model
    from bdm_amd.sampling import batch_streams, bdm_blending
    from bdm_amd.utils.procedural import fill_module_
    dev = torch.device("cuda", 0)
    cfg = ProjectConfig()
    cfg.dataset.max_points, cfg.aux_run.roll_step, cfg.aux_run.milestones, cfg.run.rng = 4096, 16, [1000, 968, 936], "per_shape"
    model = fill_module_(get_model(cfg).eval(), seed=cfg.run.seed).to(dev)
    with pytest.warns(UserWarning, match="PROCEDURAL"):
        pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, dev)
    batch = next(iter(SyntheticShapes(list(range(16)), 16, seed=cfg.run.seed, image_size=224, num_points=4096))).to(dev)

    def run(mode):
        monkeypatch.setattr(M, "TAPE_STEPS", mode)
        model._cond_cache = None  # (bench.py does this per trajectory: the image encoder runs inside the timed region)
        return bdm_blending(None, batch, cfg, model, pvd, streams=batch_streams(cfg, batch, dev, sample_idx=1)).points_padded().clone()

    eager = run("0")
    for _ in range(2):
        taped = run("auto")
        g = model._tape_cache
        assert g["tape"] is not None and g["off"] is None and g["tape"].python_entries <= 4
        assert torch.isfinite(taped).all() and torch.equal(taped, eager)
    # New images in the same batch object: the conditioning image and the hoisted maps are rewritten in place
    # (model.conditioning_image), so the SAME recorded step serves them -- and must give what the eager loop gives for them
    first_tape = model._tape_cache["tape"]
    batch.image_rgb.copy_(torch.rand(batch.image_rgb.shape, generator=torch.Generator().manual_seed(5)).to(dev))
    taped = run("auto")
    assert model._tape_cache["tape"] is first_tape, "a new image batch of the same shape re-recorded the step"
    other = run("0")
    assert torch.equal(taped, other) and not torch.equal(other, eager)
