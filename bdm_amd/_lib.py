"""ctypes loader for libbdm_hip.so (the C ABI declared in include/bdm_hip.h).

There is NO CPU fallback: if the shared object is missing, or a call is made with tensors
that do not live on a HIP device, an exception is raised.  PyTorch is used for device
memory and streams only; every kernel on the denoiser path lives in the library.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("BDM_LIB_PATH") or os.path.join(_HERE, "libbdm_hip.so")  # BDM_LIB_PATH: A/B timing of two builds
CSRC = os.path.join(_HERE, "csrc")
_lib = None
_keep = None  # list while tape.record() is active


class BdmHipError(RuntimeError):
    pass


def build(force: bool = False, jobs: int = 4) -> str:
    """Compile csrc/*.hip for gfx950 into bdm_amd/libbdm_hip.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", CSRC, f"-j{jobs}"]
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(args)
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise BdmHipError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C bdm_amd/csrc`). There is no CPU fallback for the HIP path.")
        handle = ctypes.CDLL(SO_PATH)
        for name, (restype, argtypes) in abi_signatures(experimental=hasattr(handle, "bdm_conv3d_3x3x3")).items():
            fn = getattr(handle, name)  # AttributeError here = header and library disagree: fail loudly
            fn.restype, fn.argtypes = restype, argtypes
        _lib = handle
    return _lib


HEADER = os.path.join(os.path.dirname(_HERE), "include", "bdm_hip.h")
_CTYPES = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
           "unsigned long long": ctypes.c_ulonglong, "unsigned int": ctypes.c_uint, "double": ctypes.c_double, "int64_t": ctypes.c_int64}


def has_experimental() -> bool:
    """Was the library built with `make EXPERIMENTAL=1` (fp32-MFMA convolution / flash attention, one-kernel sparse convolution)?"""
    return hasattr(lib(), "bdm_conv3d_3x3x3")


def experimental(name: str):
    """Entry point of the experimental build, or a clear error (never a silent fallback)."""
    fn = getattr(lib(), name, None)
    if fn is None:
        raise BdmHipError(f"{name} belongs to the experimental kernel families; rebuild with `make -C bdm_amd/csrc EXPERIMENTAL=1`")
    return fn


def abi_signatures(header: str = HEADER, experimental: bool = False):
    """{function: (restype, [argtypes])} parsed from include/bdm_hip.h, the single source of truth for the C ABI:
    every exported function gets its ctypes prototype from its declaration (a size_t result or a long long stride
    is never squeezed through ctypes' default 32-bit int)."""
    import re
    text = open(header).read()
    if not experimental:
        text = re.sub(r"#ifdef BDM_EXPERIMENTAL.*?#endif /\* BDM_EXPERIMENTAL \*/", " ", text, flags=re.S)
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    sigs = {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w \*]*?)\b(bdm_\w+)\s*\(([^)]*)\)\s*;", text):
        ret = " ".join(ret.replace("const", " ").split())
        restype = ctypes.c_char_p if ret == "char *" else (ctypes.c_void_p if ret.endswith("*") else (None if ret == "void" else _CTYPES[ret]))
        argtypes = []
        for a in [a.strip() for a in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
                continue
            ty = " ".join(a.replace("const", " ").split()[:-1])  # drop the parameter name
            argtypes.append(_CTYPES[ty])
        sigs[name] = (restype, argtypes)
    return sigs


_DEBUG_SYNC = os.environ.get("BDM_DEBUG_SYNC")  # path of a breadcrumb file: serialise every call, remember the last one


def check(rc: int, what: str = ""):
    if _DEBUG_SYNC:
        with open(_DEBUG_SYNC, "w") as f:
            f.write(what)
        torch.cuda.synchronize()
    if rc != 0:
        msg = lib().bdm_last_error()
        raise BdmHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a tensor as a plain int (None -> NULL).  Refuses host tensors: no CPU path.  Every function has its
    argtypes declared from the header (lib()), so ints and None convert to void* without a ctypes object per argument --
    with ~1300 pointer arguments per denoiser forward that object construction was a measurable part of the enqueue time."""
    if t is None:
        return None
    if not t.is_cuda:
        raise BdmHipError("bdm_amd operators run on a HIP device only; got a CPU tensor (no CPU fallback)")
    if _keep is not None:  # a launch tape is being recorded (tape.py): it owns every buffer whose address it holds
        _keep.append(t)
    return t.data_ptr()


def stream():
    """hipStream_t of torch's current stream on the current device (the raw C accessors: `torch.cuda.current_stream()`
    costs ~8 us of Python per call, which at ~300 launches per denoiser forward was a quarter of the enqueue time)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def f32(t):
    if t.dtype != torch.float32:
        raise BdmHipError(f"expected float32, got {t.dtype}")
    return t.contiguous()


def i32(t):
    return (t if t.dtype == torch.int32 else t.int()).contiguous()


# argument "casts" kept for readability at the call sites; the declared argtypes do the conversion
c_float = float
c_int = int
c_ll = int
