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
