"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/bdm_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(experimental=False):
    """Functions the header declares; the `#ifdef BDM_EXPERIMENTAL` blocks (superseded kernel families, `make EXPERIMENTAL=1`) only
    on request."""
    text = open(os.path.join(ROOT, "include", "bdm_hip.h")).read()
    if not experimental:
        text = re.sub(r"#ifdef BDM_EXPERIMENTAL.*?#endif /\* BDM_EXPERIMENTAL \*/", "", text, flags=re.S)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bdm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bdm_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.SO_PATH)
    is_experimental = hasattr(lib, "bdm_conv3d_3x3x3")
    names = declared_symbols(is_experimental)
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in bdm_hip.h but not exported: {missing}"
    if not is_experimental:  # the default library holds ONE kernel family per operator (+ the bf16x6 fallback): nothing experimental
        extra = [n for n in declared_symbols(True) if n not in names and hasattr(lib, n)]
        assert not extra, f"experimental entry points in the default build: {extra}"
    lib.bdm_abi_version.restype = ctypes.c_int
    assert lib.bdm_abi_version() >= 1


def test_every_declared_symbol_has_a_ctypes_prototype():
    """_lib.lib() declares restype + argtypes for every function of the header (no 32-bit default int for size_t
    results or long long strides)."""
    from bdm_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    sigs = _lib.abi_signatures(experimental=_lib.has_experimental())
    assert sorted(sigs) == declared_symbols(_lib.has_experimental())
    lib = _lib.lib()
    for name, (restype, argtypes) in sigs.items():
        fn = getattr(lib, name)
        assert fn.restype is restype and list(fn.argtypes) == argtypes, name
    assert lib.bdm_rasterize_workspace_bytes.restype is ctypes.c_size_t
    assert lib.bdm_group_norm_workspace_bytes.restype is ctypes.c_size_t
    assert ctypes.c_longlong in lib.bdm_pointwise_conv.argtypes


def test_no_cpu_fallback():
    """Host tensors are refused: the product path must not silently run on the CPU."""
    import pytest
    import torch
    from bdm_amd import _lib
    with pytest.raises(_lib.BdmHipError):
        _lib.ptr(torch.zeros(3))
    from bdm_amd.functional import _backend
    with pytest.raises(RuntimeError):
        _backend.ball_query(torch.zeros(1, 3, 4), torch.zeros(1, 3, 8), 0.1, 4)


def test_product_does_not_import_oracle():
    """Nothing under bdm_amd/ may import, link or execute the oracle."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "bdm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|from\s+\.\.?oracle|liboracle", src, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_pvcnn_backend_extension_module_has_the_reference_surface():
    """The compiled torch extension `_pvcnn_backend` (bdm_amd/csrc/torch_binding.cpp) exposes the 12 names of the
    reference's bindings.cpp:10-37 and rejects CPU tensors with a RuntimeError as the reference's CHECK_CUDA does."""
    import importlib
    import sys
    import pytest
    import torch
    from bdm_amd import _lib
    if not os.path.exists(os.path.join(os.path.dirname(_lib.SO_PATH), "_pvcnn_backend.so")):
        _lib.build()
    sys.path.insert(0, os.path.dirname(_lib.SO_PATH))
    try:
        mod = importlib.import_module("_pvcnn_backend")
    finally:
        sys.path.pop(0)
    names = {"gather_features_forward", "gather_features_backward", "furthest_point_sampling", "ball_query", "grouping_forward",
             "grouping_backward", "three_nearest_neighbors_interpolate_forward", "three_nearest_neighbors_interpolate_backward",
             "trilinear_devoxelize_forward", "trilinear_devoxelize_backward", "avg_voxelize_forward", "avg_voxelize_backward"}
    assert names <= set(dir(mod))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        mod.ball_query(torch.zeros(1, 3, 4), torch.zeros(1, 3, 8), 0.1, 4)
    with pytest.raises(RuntimeError):
        mod.furthest_point_sampling(torch.zeros(1, 3, 4), 2)
