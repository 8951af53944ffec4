import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: needs a real MI355X AND minutes of CPU-oracle time; only selected by -m gpu_slow")
    import torch
    torch.set_num_threads(_usable_cpus())  # the oracle runs on the host: do not oversubscribe a cgroup-limited box


@pytest.fixture(scope="session")
def oracle_ops():
    from oracle import ops
    ops.build()
    return ops


@pytest.fixture(scope="session")
def hip():
    """The HIP backend; fails loudly (never falls back) if the extension or the GPU is missing."""
    import torch
    from bdm_amd import _lib
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    _lib.lib()
    from bdm_amd.functional import _backend
    return _backend


# ---- GPU suite packaging ------------------------------------------------------------------------------
# * `gpu_slow` = the live-oracle forms of the full-length trajectory tests (minutes of HOST time each): never part of
#   `-m gpu` or `-m "not gpu"`; run them with `-m gpu_slow`.
# * `-m gpu` runs the cheapest, most strongly pinned files first (network and module goldens from the reference's own classes,
#   the seven operators, the Philox streams), the long trajectories last: a time limit cuts the tail, not the pins.
# * tests/durations.json = seconds per test recorded on the builder's GPU box (`tools/record_durations.py`); the driver's host
#   was measured 1.8x slower on host-bound tests, so the collection FAILS when recorded x 1.8 exceeds the limit.
GPU_FILE_ORDER = ["test_hip_net.py", "test_module_goldens.py", "test_hip_ops.py", "test_rng.py", "test_abi.py", "test_hip_dense.py",
                  "test_hip_small_glue.py", "test_hip_sampler.py", "test_backward_ops.py", "test_hip_bench_variants.py", "test_hip_uninit.py",
                  "test_hip_trajectory.py", "test_hip_teacher_forced.py", "test_hip_cli.py", "test_hip_full_size.py",
                  "test_hip_full_trajectory.py"]
GPU_SUITE_LIMIT_S, SLOW_HOST_FACTOR = 600.0, 1.8


def _recorded_durations():
    import json
    try:
        return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "durations.json")))
    except OSError:
        return {}


def pytest_collection_modifyitems(config, items):
    expr = config.option.markexpr or ""
    if "gpu_slow" not in expr:
        slow = [it for it in items if it.get_closest_marker("gpu_slow")]
        if slow:
            config.hook.pytest_deselected(items=slow)
            items[:] = [it for it in items if not it.get_closest_marker("gpu_slow")]
    rank = {name: i for i, name in enumerate(GPU_FILE_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(rank)))   # stable: file order inside a file kept
    if expr.strip() == "gpu":
        rec = _recorded_durations().get("tests", {})
        selected = [it for it in items if it.get_closest_marker("gpu")]
        total = sum(rec.get(it.nodeid, 0.0) for it in selected)
        if total * SLOW_HOST_FACTOR > GPU_SUITE_LIMIT_S:
            raise pytest.UsageError(f"GPU suite over budget: {total:.0f} s recorded x {SLOW_HOST_FACTOR} > {GPU_SUITE_LIMIT_S:.0f} s "
                                    "(tests/durations.json): move host-bound oracle work into fixtures or under gpu_slow")


# BDM_RECORD_DURATIONS=<path>: write {"tests": {nodeid: seconds (setup + call + teardown)}, ...} at the end of the session.
# On the GPU box:  BDM_RECORD_DURATIONS=gpurun_out/durations.json python -m pytest tests -m gpu -q ; then copy to tests/durations.json
_durations = {}


def pytest_runtest_logreport(report):
    if os.environ.get("BDM_RECORD_DURATIONS"):
        _durations[report.nodeid] = _durations.get(report.nodeid, 0.0) + float(report.duration)


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("BDM_RECORD_DURATIONS")
    if path and _durations:
        import json
        import platform
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        meta = {"host_cpus": _usable_cpus(), "host": platform.processor() or platform.machine(), "exitstatus": int(exitstatus),
                "total_s": round(sum(_durations.values()), 1)}
        json.dump({"meta": meta, "tests": {k: round(v, 2) for k, v in sorted(_durations.items())}}, open(path, "w"), indent=0)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line per parity figure the session measured (helpers.parity): `name  value / bound  (fraction of the bound)`, worst
    fraction first, so that the tail of a `-q` run shows the MARGIN of every parity claim, not only that it held."""
    try:
        import helpers
    except ImportError:
        return
    rows = list(helpers.PARITY)
    if not rows:
        return
    tr = terminalreporter
    tr.write_sep("=", f"parity figures ({len(rows)}): measured / asserted bound")
    worst = {}
    for name, value, bound, note in rows:   # a name reported several times (parametrised steps): keep its worst
        if name not in worst or value / max(bound, 1e-300) > worst[name][0] / max(worst[name][1], 1e-300):
            worst[name] = (value, bound, note)
    ordered = sorted(worst.items(), key=lambda kv: -(kv[1][0] / max(kv[1][1], 1e-300)))
    headline = [kv for kv in ordered if kv[0].startswith("traj_") and " segment " not in kv[0]]
    segs = sorted(kv for kv in ordered if kv[0].startswith("traj_") and " segment " in kv[0])
    others = [kv for kv in ordered if not kv[0].startswith("traj_")]
    by_traj = {}
    for name, (value, bound, note) in segs:   # one line per trajectory: its error at the end of every schedule segment
        by_traj.setdefault(name.split(" segment ")[0], []).append(value)
    for tname, vals in by_traj.items():
        tr.write_line(f"PARITY {tname} per schedule segment: " + " ".join(f"{v:.1e}" for v in vals))
    for name, (value, bound, note) in others[:60][::-1] + headline[::-1]:   # headline trajectories last = nearest the tail
        frac = value / max(bound, 1e-300)
        tr.write_line(f"PARITY {name:<58s} {value:9.3e} / {bound:7.1e}  ({frac:5.1%} of bound){'  ' + note if note else ''}")
    if len(others) > 60:
        tr.write_line(f"PARITY ... {len(others) - 60} more figures below {others[60][1][0] / max(others[60][1][1], 1e-300):.1%} of their bounds")
