"""Do the known-answer cases of tests/test_oracle_ops.py really pin the oracle's documented rules?
For each rule a deliberately WRONG variant of oracle/pvcnn_ops_ref.c is built (-DORACLE_MUTATION=k: non-strict radius
test, missing first-hit fill, FPS without the 512-lane tie rule, '>=' / '<=' in the FPS reductions, '<=' 3-NN cascades,
missing 3-NN clamp, reversed voxel summation order, devoxelisation that always addresses the upper neighbour) and the very
same test functions are run against it: at least one must fail.  The unmutated build passes all of them."""
import ctypes
import os
import subprocess

import pytest

import test_oracle_ops as KA

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [getattr(KA, n) for n in dir(KA) if n.startswith("test_")]
MUTATIONS = {
    1: "ball query: d2 <= r2 instead of the strict d2 < r2 (ball_query.cu:39)",
    2: "ball query: unused slots not filled with the first hit (ball_query.cu:40-44)",
    3: "FPS: plain first-index arg-max, no (k mod 512, k) rule (sampling.cu:120-160)",
    4: "FPS stage 1: '>=' keeps the LAST of equal candidates in a lane (sampling.cu:128-131)",
    5: "FPS stage 2: right lane replaces the left on ties (sampling.cu:154)",
    6: "3-NN: '<=' cascades keep the later centre on ties (neighbor_interpolate.cu:45-58)",
    7: "3-NN: no 1e-10 lower clamp of the squared distances (neighbor_interpolate.cu:61-63)",
    8: "avg_voxelize: descending point order inside a voxel (the oracle's declared order is ascending)",
    9: "devoxelize: the +1 neighbour is addressed even when frac == 0 (trilinear_devox.cu:64-75)",
}


def _build(tmp_path, k):
    so = tmp_path / f"liboracle_mut{k}.so"
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-std=c11", f"-DORACLE_MUTATION={k}",
                           "-shared", "-o", str(so), os.path.join(ROOT, "oracle", "pvcnn_ops_ref.c"), "-lm"])
    return ctypes.CDLL(str(so))


def _failures(ops):
    failed = []
    for case in CASES:
        try:
            case(ops)
        except AssertionError:
            failed.append(case.__name__)
    return failed


def test_unmutated_build_passes(oracle_ops, tmp_path, monkeypatch):
    monkeypatch.setattr(oracle_ops, "_lib", _build(tmp_path, 0))
    assert _failures(oracle_ops) == []


@pytest.mark.parametrize("k", sorted(MUTATIONS))
def test_known_answers_reject_the_mutant(oracle_ops, tmp_path, monkeypatch, k):
    monkeypatch.setattr(oracle_ops, "_lib", _build(tmp_path, k))
    failed = _failures(oracle_ops)
    print(f"mutation {k} ({MUTATIONS[k]}): rejected by {failed}")
    assert failed, f"no known-answer case notices: {MUTATIONS[k]}"
