"""Full-length coupled trajectories, HIP path vs CPU oracle, held to the LITERAL north-star bound: relative L2 <= 1e-3 on the
final cloud, no self-sensitivity escape hatch.

Schedule: the recipes' real milestones [1000, 968, 936, 872, 128, 64, 32, 0], roll_step 16 -- 1000 PC^2 + 80 PVD forwards and
5 blends (Blending), 995 PC^2 + 75 PVD + 5 fused steps (Merging) -- at N = 1024, B = 1, projection conditioning at every
step, every random draw injected identically on both sides.

Weights: procedural, with the denoisers' LAST layer scaled by HEAD_SCALE.  The reference itself initialises that layer with
N(0, 1e-6) (experiments/model/point_cloud_model.py:38-39), i.e. its own fresh model is a near-zero noise predictor; with
HEAD_SCALE = 1 the procedural network is chaotic (the ORACLE moved by one ulp ends 2.6e-2 away from itself after only 100
steps).  HEAD_SCALE below is the largest of {1, 0.3, 0.1, 0.03, 0.01} at which the oracle's own 1-ulp self-sensitivity over
the full 1080-forward schedule stays < 1e-4 (tools/chaos_probe.py; the measured table is in DESIGN.md section 5), so that a
pass/fail at 1e-3 is a statement about the kernels and not about chaos.  Every layer below the head runs at full scale:
the per-step teacher-forced tests (tests/test_hip_teacher_forced.py) cover the head at scale 1.
"""
import pytest
import torch

from helpers import first_segment_past, golden_trajectory, parity, rel_l2
import trajectory_case as case

HEAD_SCALE = 0.1
NORTH_STAR = 1e-3


# Early-warning lines of the CALM twins (head scale 0.03; the 0.1 fixtures keep the literal bound only: their figures moved between 2e-6
# and 3e-4 from one summation order of a kernel to the next in rounds 4 - 5 -- chaos of the procedural network)
MARGIN_LINE = {"blending_n1024_h003": 1e-4, "merging_n1024_h003": 1e-4}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["blending_n1024_h003", "merging_n1024_h003", "blending_n1024", "merging_n1024"])
def test_full_trajectory_vs_oracle_fixture(hip, name):
    """The default (`-m gpu`) form: the oracle's final cloud comes from tests/golden/traj_<name>.npz, written by
    oracle/gen_golden_traj.py (the same case, the same oracle, run once in the build container: ~3 min of host time each that the
    GPU suite no longer pays per run).  The HIP side is the full 1080- / 1075-forward trajectory, nothing shortened."""
    g = golden_trajectory(name)
    c = case.build(int(g["N"]), head_scale=float(g["head_scale"]), merging=bool(g["merging"]), B=int(g["B"]))
    assert list(g["milestones"]) == list(c.milestones) and int(g["roll_step"]) == c.roll_step
    assert len(case.program_order(c.milestones, c.roll_step, c.merging)) == int(g["forwards"])
    with case.segments() as seg:
        got = case.run_hip(c)
    err = parity(f"traj_{name} final cloud, {int(g['forwards'])} forwards", rel_l2(got, torch.from_numpy(g["final"])), NORTH_STAR)
    first, curve = first_segment_past(NORTH_STAR, seg.clouds, g)
    for i, e in enumerate(curve):
        parity(f"traj_{name} segment {i}", e, NORTH_STAR)
    print(f"full trajectory {name} ({int(g['forwards'])} forwards, N={int(g['N'])}) vs the oracle fixture: final rel-L2 {err:.3e}; "
          "per segment " + " ".join(f"{e:.1e}" for e in curve))
    assert err <= NORTH_STAR, (f"final rel-L2 {err:.3e} > {NORTH_STAR}; first schedule segment past the bound: {first} "
                               f"(segment curve {['%.2e' % e for e in curve]})")
    line = MARGIN_LINE.get(name)
    if line is not None:
        assert err <= line, (f"margin gone: final rel-L2 {err:.3e} is inside the 1e-3 bound but past the {line:.0e} early-warning line of the calm "
                             f"(head 0.03) fixture; segment curve {['%.2e' % e for e in curve]}")


@pytest.mark.gpu_slow
def test_full_blending_trajectory_literal_bound(hip, oracle_ops):
    c = case.build(1024, head_scale=HEAD_SCALE, merging=False)
    assert len(case.program_order(c.milestones, c.roll_step)) == 1080
    ref = case.run_oracle(c)
    got = case.run_hip(c)
    err = parity("traj_live_blending_n1024 final cloud", rel_l2(got, ref), NORTH_STAR)
    print(f"full BDM-Blending trajectory (1000 PC^2 + 80 PVD + 5 blends, N=1024): final rel-L2 {err:.3e}")
    assert err <= NORTH_STAR


@pytest.mark.gpu_slow
def test_full_merging_trajectory_literal_bound(hip, oracle_ops):
    c = case.build(1024, head_scale=HEAD_SCALE, merging=True)
    order = case.program_order(c.milestones, c.roll_step, merging=True)
    assert sum(k in ("recon", "branch") for k, _ in order) == 995 and sum(k == "prior" for k, _ in order) == 75
    assert sum(k == "fuse" for k, _ in order) == 5
    ref = case.run_oracle(c)
    got = case.run_hip(c)
    err = parity("traj_live_merging_n1024 final cloud", rel_l2(got, ref), NORTH_STAR)
    print(f"full BDM-Merging trajectory (995 PC^2 + 75 PVD + 5 fused, N=1024): final rel-L2 {err:.3e}")
    assert err <= NORTH_STAR
