"""Launch tape: one reverse step recorded as a flat list of C-ABI calls, replayed without the Python module code.

Why: a single small shape (config C1: B=1, N=1024) is bound by the HOST -- ~226 dependent launches per step, each
costing 14-16 us of Python (module forward, shape checks, output allocation, argument conversion) for a kernel that
runs 4-6 us.  A hipGraph does not help on ROCm 7.2 (its replay walks the nodes on a runtime thread at ~15 us per
node: tools/time_loop.py).  All arguments of a step are static once the step runs on fixed buffers (the timestep,
the five scheduler scalars and the noise live in device memory: model._step_buffers), so the step is recorded ONCE
while it executes eagerly:

  * every `_lib.lib().bdm_*(...)` call -> (ctypes function, argument tuple)   [recorded by a proxy on `_lib._lib`]
  * every tensor whose address went into such a call is kept alive by the tape (`_lib.ptr` hook), so the caching
    allocator can never hand the address to anyone else: replaying the list writes the very same buffers;
  * the few torch operators inside the step (zero-fills of the voxel-plan counters, a copy) are caught by a
    TorchDispatchMode and recorded as in-place operations on their (kept) outputs;
  * stream / event edges of the side streams go through `wait_stream` / `record_event` / `wait_event` below.

Replay = `for fn, args in calls: fn(*args)`: ~3 us of host time per launch.  Same kernels, same arguments, same
order, same streams as the eager step: the results are bit-identical (tests/test_hip_trajectory.py).
An operator the recorder does not understand marks the tape broken and the caller stays on the eager path.
"""
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib as L

_active = None  # the LaunchTape being recorded (None: not recording)


def _host_only(name):
    return (not name.startswith("bdm_") or name.endswith("_bytes") or name.endswith("_elems") or name.endswith("_slices")
            or name in ("bdm_last_error", "bdm_abi_version"))


class _RecordingLib:
    """Stands in for the ctypes handle while a step is recorded: calls go through AND onto the tape."""

    def __init__(self, handle, tape):
        self._h, self._tape = handle, tape

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if _host_only(name):
            return fn
        calls = self._tape.calls

        def call(*args):
            calls.append((fn, args))
            return fn(*args)
        self.__dict__[name] = call
        return call


def _tensors(tree, out):
    if isinstance(tree, torch.Tensor):
        out.append(tree)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            _tensors(v, out)
    elif isinstance(tree, dict):
        for v in tree.values():
            _tensors(v, out)
    return out


_ALLOC_ONLY = {"aten::empty", "aten::empty_strided", "aten::empty_like", "aten::new_empty", "aten::new_empty_strided",
               "aten::resize_", "aten::set_", "aten::detach", "aten::alias", "aten::lift_fresh", "aten::_unsafe_view"}
_FILLS = {"aten::zeros": 0, "aten::zeros_like": 0, "aten::new_zeros": 0, "aten::ones": 1, "aten::ones_like": 1, "aten::new_ones": 1}


def _py(stream, fn, args, kwargs):
    """A torch operator on the tape: re-issued on the stream it was recorded on; returns 0 like the C functions."""
    if stream is None:
        fn(*args, **kwargs)
    else:
        with torch.cuda.stream(stream):
            fn(*args, **kwargs)
    return 0


def _into(stream, out, fn, args, kwargs):
    return _py(stream, lambda: out.copy_(fn(*args, **kwargs)), (), {})


class _TorchOps(TorchDispatchMode):
    def __init__(self, tape):
        super().__init__()
        self.tape = tape

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        tape = self.tape
        touched = _tensors((args, kwargs, out), [])
        if not any(t.is_cuda for t in touched):
            return out
        name = func._schema.name
        if func.is_view or name in _ALLOC_ONLY:
            return out
        if any(not t.is_cuda and t.numel() > 0 for t in touched) or name == "aten::_local_scalar_dense":
            tape.broken = f"{name}: host <-> device traffic inside the step"
            return out
        cur = torch.cuda.current_stream()
        stream = None if cur == tape.main_stream else cur
        tape.keep.extend(touched)
        if func._schema.is_mutable:  # in-place / out= operator: the same call lands in the same (kept) tensors
            tape.calls.append((_py, (stream, func, args, kwargs)))
        elif name in _FILLS and isinstance(out, torch.Tensor):
            tape.calls.append((_py, (stream, out.fill_, (_FILLS[name],), {})))
        elif name in ("aten::full", "aten::full_like", "aten::new_full") and isinstance(out, torch.Tensor):
            value = kwargs["fill_value"] if "fill_value" in kwargs else args[2 if name == "aten::new_full" else 1]
            tape.calls.append((_py, (stream, out.fill_, (value,), {})))
        elif isinstance(out, torch.Tensor):  # functional operator: recompute, then land in the recorded output
            tape.calls.append((_into, (stream, out, func, args, kwargs)))
        else:
            tape.broken = f"{name}: operator with a non-tensor result"
        tape.torch_ops.append(name)
        return out


class LaunchTape:
    def __init__(self):
        self.calls, self.keep, self.torch_ops, self.broken = [], [], [], None
        self.main_stream = None

    def __len__(self):
        return len(self.calls)

    def replay(self):
        for fn, args in self.calls:
            rc = fn(*args)
            if rc:
                msg = L.lib().bdm_last_error()
                raise L.BdmHipError(f"launch tape: {getattr(fn, '__name__', fn)} failed (code {rc}): {msg.decode() if msg else ''}")


class record:
    """`with record() as tape:` -- run one step eagerly and record it.  Check `tape.broken` afterwards."""

    def __enter__(self):
        global _active
        if _active is not None:
            raise RuntimeError("launch tape: recordings do not nest")
        self.tape = LaunchTape()
        self.tape.main_stream = torch.cuda.current_stream()
        self._saved = L.lib()
        L._lib = _RecordingLib(self._saved, self.tape)
        L._keep = self.tape.keep
        self._mode = _TorchOps(self.tape)
        self._mode.__enter__()
        _active = self.tape
        return self.tape

    def __exit__(self, et, ev, tb):
        global _active
        _active = None
        self._mode.__exit__(et, ev, tb)
        L._keep = None
        L._lib = self._saved
        if et is not None:
            self.tape.broken = f"exception while recording: {ev!r}"
        return False


# ---- stream / event edges (the side streams of pvcnn.plan_sampling_chain and PVConv's point branch) -----------------------
def wait_stream(waiter, other):
    waiter.wait_stream(other)
    if _active is not None:
        _active.calls.append((_py, (None, waiter.wait_stream, (other,), {})))


def record_event(event, stream):
    event.record(stream)
    if _active is not None:
        _active.calls.append((_py, (None, event.record, (stream,), {})))


def wait_event(event):
    """The CURRENT stream waits for `event` (torch's `event.wait()`)."""
    s = torch.cuda.current_stream()
    s.wait_event(event)
    if _active is not None:
        _active.calls.append((_py, (None, s.wait_event, (event,), {})))
