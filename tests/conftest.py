import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
