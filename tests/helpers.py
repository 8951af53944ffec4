import numpy as np
import pytest
import torch

# The superseded kernel families (fp32-MFMA convolution / flash attention, one-kernel sparse convolution) live in
# bdm_amd/csrc/experimental and are built by `make EXPERIMENTAL=1` only; tests of them run against such a build:
#   make -C bdm_amd/csrc EXPERIMENTAL=1 OUT=../libbdm_hip_experimental.so EXT=/tmp/_pvcnn_backend_exp.so
#   BDM_LIB_PATH=bdm_amd/libbdm_hip_experimental.so python -m pytest tests -m gpu
experimental = pytest.mark.skipif("not __import__('bdm_amd._lib', fromlist=['x']).has_experimental()",
                                  reason="kernel family of the EXPERIMENTAL=1 build")


def seeded(shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))


def point_cloud_inputs(B, C, N, seed):
    """Same draw as oracle/gen_golden.py: xyz ~ 0.5*N(0,1), features ~ N(0,1)."""
    x = seeded((B, C, N), seed)
    x[:, :3] *= 0.5
    return x


def rel_l2(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ---- parity figures for the terminal summary (tests/conftest.py::pytest_terminal_summary) -------------------------------------
# `-q` swallows the tests' prints, so a passing run would only prove "<= bound", not how far from it.  Parity tests report their
# measured figure here; the summary prints one line per figure (`name  value / bound`) at the end of the session, where the
# driver's record of the run keeps it.
PARITY = []


def parity(name, value, bound, note=""):
    """Record `value` (measured) against `bound` (asserted by the caller) under `name`; returns the value."""
    PARITY.append((str(name), float(value), float(bound), str(note)))
    return float(value)


def current_test():
    """`file::test[param]` of the running test (pytest sets PYTEST_CURRENT_TEST), without the directory and the phase."""
    import os
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].replace("tests/", "")


def check_rel_l2(got, ref, tol, what=""):
    """assert rel_l2(got, ref) < tol, and put the figure on the session's parity record under the test's own name."""
    v = parity(current_test() + (" " + what if what else ""), rel_l2(got, ref), tol)
    assert v < tol, f"rel-L2 {v:.3e} >= {tol:.1e} {what}"
    return v


def first_segment_past(bound, got_segments, golden, key="segment_{}"):
    """Index and error of the first schedule segment whose end-of-segment cloud is further than `bound` from the oracle fixture's
    (tests/golden/traj_*.npz hold `segment_0` .. `segment_6`), plus the whole curve; (None, curve) when none is."""
    curve = []
    for i, x in enumerate(got_segments):
        k = key.format(i)
        if k not in golden:
            break
        curve.append(rel_l2(x, torch.from_numpy(golden[k])))
    first = next((i for i, e in enumerate(curve) if e > bound), None)
    return first, curve


def golden_trajectory(name):
    """tests/golden/traj_<name>.npz (oracle/gen_golden_traj.py): case description + the oracle's final cloud."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"traj_{name}.npz"))
    return {k: g[k] for k in g.files}


def oracle_spread(name):
    """Oracle-vs-oracle yardstick of a trajectory fixture: the relative L2 distance between the final clouds of the
    fixture of record and of the SAME case run by the SAME oracle at another reduction order (`traj_<name>_alt*.npz`: another
    `torch.set_num_threads`, oracle/gen_golden_traj.py --threads) -- the largest over the alternates present.  Owes nothing to the
    product: both sides are oracle/.  None when the fixture has no alternate."""
    import glob
    import os
    g = golden_trajectory(name)
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    alts = sorted(glob.glob(os.path.join(here, f"traj_{name}_alt*.npz")))
    if not alts:
        return None
    ref = torch.from_numpy(g["final"])
    return max(rel_l2(torch.from_numpy(np.load(a)["final"]), ref) for a in alts)
