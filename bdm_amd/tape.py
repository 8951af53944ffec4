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

Replay: the recorded list is handed to the C-side step executor of the library (include/bdm_hip.h section 5, csrc/tape.hip):
`LaunchTape.finalize()` appends every C-ABI call (name + 8-byte argument slots), every zero-fill / device copy (memset /
memcpy) and every stream / event edge to a `bdm_tape`, and `replay()` is then ONE ctypes call per run of native entries --
Python touches the step once, not once per launch.  Anything the executor has no native form for (a torch operator other than
a fill or a plain copy) stays a Python entry between two native runs.  `BDM_TAPE_NATIVE=0` keeps the pure-Python replay
(`for fn, args in calls: fn(*args)`, ~3 us of interpreter time per launch) for A/B timing.  Same kernels, same arguments, same
order, same streams as the eager step: the results are bit-identical (tests/test_hip_trajectory.py).
An operator the recorder does not understand marks the tape broken and the caller stays on the eager path.
"""
import ctypes
import os
import struct
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib as L

_active = None  # the LaunchTape being recorded (None: not recording)


def _host_only(name):
    # (bdm_tape_*: the executor's own management calls -- e.g. another tape's __del__ running during a recording -- are never steps)
    return (not name.startswith("bdm_") or name.endswith("_bytes") or name.endswith("_elems") or name.endswith("_slices")
            or name.startswith("bdm_tape_") or name in ("bdm_last_error", "bdm_abi_version"))


# BDM_TAPE_SKIP=bdm_x,bdm_y: TIMING EXPERIMENTS ONLY -- these C-ABI functions run while the step is recorded but are not put on the tape,
# so the replayed step shows what the loop would cost WITHOUT their launches (results are garbage: their outputs go stale).  The way a
# fusion is priced before it is built (tools/replay_host_time.py).
_PRICE_SKIP = frozenset(v for v in os.environ.get("BDM_TAPE_SKIP", "").split(",") if v)


class _RecordingLib:
    """Stands in for the ctypes handle while a step is recorded: calls go through AND onto the tape."""

    def __init__(self, handle, tape):
        self._h, self._tape = handle, tape

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if _host_only(name):
            return fn
        calls = self._tape.calls
        if name in _PRICE_SKIP:   # pricing experiment (BDM_TAPE_SKIP): executed while recording, LEFT OUT of the replayed step
            self.__dict__[name] = fn
            return fn

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
    if stream is None or torch.cuda.current_stream() == stream:
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
        if tape.pool is not None:
            # storages that appear as an operator's result without being one of its inputs were allocated during the recording,
            # i.e. inside the tape's private pool (views share their base's storage and so inherit its status)
            ins = {t.untyped_storage().data_ptr() for t in _tensors((args, kwargs), []) if t.is_cuda}
            for t in _tensors(out, []):
                if t.is_cuda and t.untyped_storage().data_ptr() not in ins:
                    tape.pool_storages.add(t.untyped_storage().data_ptr())
        name = func._schema.name
        if func.is_view or name in _ALLOC_ONLY:
            return out
        if any(not t.is_cuda and t.numel() > 0 for t in touched) or name == "aten::_local_scalar_dense":
            tape.broken = f"{name}: host <-> device traffic inside the step"
            return out
        cur = torch.cuda.current_stream()
        stream = cur  # replayed on the stream it was recorded on, whatever stream is current then
        if tape.pool is None:
            tape.keep.extend(touched)
        raw = cur.cuda_stream
        if func._schema.is_mutable:  # in-place / out= operator: the same call lands in the same (kept) tensors
            tape.calls.append((_py, (stream, func, args, kwargs)))
            dst = args[0] if args and isinstance(args[0], torch.Tensor) else None
            if name == "aten::zero_" or (name == "aten::fill_" and len(args) > 1 and not isinstance(args[1], torch.Tensor) and args[1] == 0):
                tape.native[len(tape.calls) - 1] = ("memset", dst, raw)
            elif (name == "aten::copy_" and len(args) > 1 and isinstance(args[1], torch.Tensor) and args[1].is_cuda
                  and args[1].dtype == dst.dtype and args[1].shape == dst.shape):
                tape.native[len(tape.calls) - 1] = ("memcpy", dst, args[1], raw)
        elif name in _FILLS and isinstance(out, torch.Tensor):
            tape.calls.append((_py, (stream, out.fill_, (_FILLS[name],), {})))
            if _FILLS[name] == 0:
                tape.native[len(tape.calls) - 1] = ("memset", out, raw)
        elif name in ("aten::full", "aten::full_like", "aten::new_full") and isinstance(out, torch.Tensor):
            value = kwargs["fill_value"] if "fill_value" in kwargs else args[2 if name == "aten::new_full" else 1]
            tape.calls.append((_py, (stream, out.fill_, (value,), {})))
        elif isinstance(out, torch.Tensor):  # functional operator: recompute, then land in the recorded output
            tape.calls.append((_into, (stream, out, func, args, kwargs)))
            src = args[0] if args and isinstance(args[0], torch.Tensor) else None
            if (name in ("aten::clone", "aten::_to_copy", "aten::contiguous") and src is not None and src.is_cuda and src.dtype == out.dtype
                    and src.shape == out.shape and src.is_contiguous() and out.is_contiguous()):
                tape.native[len(tape.calls) - 1] = ("memcpy", out, src, raw)
        else:
            tape.broken = f"{name}: operator with a non-tensor result"
        tape.torch_ops.append(name)
        return out


NATIVE = os.environ.get("BDM_TAPE_NATIVE", "1") == "1"
# Recording inside a private memory pool (torch.cuda.MemPool, what graph capture uses): the step's intermediates are freed and
# re-used WITHIN the step exactly as in the eager loop (cache-hot blocks), and nobody outside the tape can be handed an address of the
# pool between replays.  Without it the tape keeps every intermediate alive (distinct buffers: measured 1 % slower per B=16 step).
POOL = hasattr(torch.cuda, "MemPool") and hasattr(torch.cuda, "use_mem_pool")  # (tools/tape_check.py --no-pool flips it)
_U64 = (1 << 64) - 1


def _slots(name, args):
    """8-byte argument slots of a recorded C-ABI call (include/bdm_hip.h section 5): integers and pointers as they are (two's
    complement), float arguments as the bit pattern of a double."""
    argtypes = L.abi_signatures(experimental=L.has_experimental())[name][1] if name not in _slots.cache else _slots.cache[name]
    _slots.cache[name] = argtypes
    if len(argtypes) != len(args):
        raise L.BdmHipError(f"launch tape: {name} recorded with {len(args)} arguments, the header declares {len(argtypes)}")
    out = []
    for ty, v in zip(argtypes, args):
        if ty in (ctypes.c_float, ctypes.c_double):
            out.append(struct.unpack("<Q", struct.pack("<d", float(v)))[0])
        elif v is None:
            out.append(0)
        elif hasattr(v, "_obj"):  # ctypes.byref(x): a host out-parameter (e.g. slices_out); x stays alive with the recorded args
            out.append(ctypes.addressof(v._obj))
        elif isinstance(v, ctypes._SimpleCData):
            out.append(int(v.value or 0) & _U64)
        else:
            out.append(int(v) & _U64)
    return out


_slots.cache = {}


class LaunchTape:
    def __init__(self):
        self.calls, self.keep, self.torch_ops, self.broken = [], [], [], None
        self.native = {}     # index into calls -> native form of a non-C-ABI entry: ("memset", tensor, raw stream) | ...
        self.main_stream = None
        self.handle, self.program, self.python_entries = None, None, None
        self.pool = None     # the private memory pool the step was recorded in (owns every address allocated while recording)
        self.pool_storages = set()   # storage addresses allocated during the recording (inside the pool)

    def __len__(self):
        return len(self.calls)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and L is not None and getattr(L, "_lib", None) is not None:  # (module globals are gone at interpreter shutdown)
            L._lib.bdm_tape_destroy(h)

    @staticmethod
    def _contiguous_bytes(t):
        return t.numel() * t.element_size() if t.is_contiguous() else None

    def finalize(self):
        """Hand the recording to the C-side executor: program = [(first, count) native runs | (fn, args) Python entries]."""
        if not NATIVE or self.handle is not None or self.broken:
            return self
        lib = L.lib()
        h = lib.bdm_tape_create()
        program, run_first, py = [], None, 0

        def close_run():
            nonlocal run_first
            n = lib.bdm_tape_length(h)
            if run_first is not None and n > run_first:
                program.append((run_first, n - run_first))
            run_first = None

        for i, (fn, args) in enumerate(self.calls):
            ok = False
            nat = self.native.get(i)
            if isinstance(fn, ctypes._CFuncPtr):
                name = fn.__name__
                sl = _slots(name, args)
                arr = (ctypes.c_ulonglong * max(len(sl), 1))(*sl)
                L.check(lib.bdm_tape_append_call(h, name.encode(), ctypes.addressof(arr), len(sl)), f"tape_append_call({name})")
                ok = True
            elif nat is not None and nat[0] == "memset":
                nb = self._contiguous_bytes(nat[1])
                if nb is not None:
                    L.check(lib.bdm_tape_append_memset(h, nat[1].data_ptr(), 0, nb, nat[2]), "tape_append_memset")
                    ok = True
            elif nat is not None and nat[0] == "memcpy":
                nb, nb2 = self._contiguous_bytes(nat[1]), self._contiguous_bytes(nat[2])
                if nb is not None and nb == nb2:
                    L.check(lib.bdm_tape_append_memcpy(h, nat[1].data_ptr(), nat[2].data_ptr(), nb, nat[3]), "tape_append_memcpy")
                    ok = True
            elif nat is not None and nat[0] == "wait_stream":
                L.check(lib.bdm_tape_append_wait_stream(h, nat[1].cuda_stream, nat[2].cuda_stream), "tape_append_wait_stream")
                ok = True
            elif nat is not None and nat[0] == "event_record":
                L.check(lib.bdm_tape_append_event_record(h, nat[1].cuda_event, nat[2].cuda_stream), "tape_append_event_record")
                ok = True
            elif nat is not None and nat[0] == "event_wait":
                L.check(lib.bdm_tape_append_event_wait(h, nat[1].cuda_stream, nat[2].cuda_event), "tape_append_event_wait")
                ok = True
            if ok:
                if run_first is None:
                    run_first = lib.bdm_tape_length(h) - 1
            else:
                close_run()
                program.append((fn, args))
                py += 1
        close_run()
        self.handle, self.program, self.python_entries = h, program, py
        return self

    def _fail(self, what, rc):
        msg = L.lib().bdm_last_error()
        raise L.BdmHipError(f"launch tape: {what} failed (code {rc}): {msg.decode() if msg else ''}")

    def replay(self):
        if self.program is not None:
            replay_c = L.lib().bdm_tape_replay
            h = self.handle
            for a, b in self.program:
                if type(a) is int:
                    rc = replay_c(h, a, b)
                    if rc:
                        self._fail(f"native entry {L.lib().bdm_tape_failed_entry(h)}", rc)
                else:
                    rc = a(*b)
                    if rc:
                        self._fail(getattr(a, "__name__", a), rc)
            return
        for fn, args in self.calls:
            rc = fn(*args)
            if rc:
                self._fail(getattr(fn, "__name__", fn), rc)


class _PinOutsiders:
    """`_lib._keep` while a step is recorded into a private pool: the pool owns every buffer allocated DURING the recording, but the
    step also bakes in addresses of longer-lived buffers allocated outside it -- `ops._ws_cache` workspaces (re-allocated when a
    larger request arrives), the packed cameras, hoisted conditioning maps, amax rings, weights packs.  Those are pinned by the
    tape, so a later eager call that replaces one of them can never free memory a recorded step still reads or writes."""

    def __init__(self, tape):
        self.tape, self.seen = tape, set()

    def append(self, t):
        sp = t.untyped_storage().data_ptr()
        if sp not in self.tape.pool_storages and sp not in self.seen:
            self.seen.add(sp)
            self.tape.keep.append(t)


class record:
    """`with record() as tape:` -- run one step eagerly and record it.  Check `tape.broken` afterwards."""

    def __enter__(self):
        global _active
        if _active is not None:
            raise RuntimeError("launch tape: recordings do not nest")
        self.tape = LaunchTape()
        self.tape.main_stream = torch.cuda.current_stream()
        self._saved = L.lib()
        base = self._saved
        while "_h" in getattr(base, "__dict__", {}):  # a profiling proxy (profiling.KernelClassProfiler): the tape holds the library's
            base = base.__dict__["_h"]                # own entry points; profiled steps are the caller's eager ones
        L._lib = _RecordingLib(base, self.tape)
        self._pool_ctx = None
        if POOL and torch.cuda.is_available():
            self.tape.pool = torch.cuda.MemPool()
            self._pool_ctx = torch.cuda.use_mem_pool(self.tape.pool)
            self._pool_ctx.__enter__()
            # the pool owns what is allocated inside the step (recycled within it, as in the eager loop); buffers from OUTSIDE are pinned
            L._keep = _PinOutsiders(self.tape)
        else:
            L._keep = self.tape.keep  # no pool: the tape owns every buffer whose address it holds
        self._mode = _TorchOps(self.tape)
        self._mode.__enter__()
        _active = self.tape
        return self.tape

    def __exit__(self, et, ev, tb):
        global _active
        _active = None
        self._mode.__exit__(et, ev, tb)
        if self._pool_ctx is not None:
            self._pool_ctx.__exit__(et, ev, tb)
        L._keep = None
        L._lib = self._saved
        if et is not None:
            self.tape.broken = f"exception while recording: {ev!r}"
        else:
            try:
                self.tape.finalize()
            except (KeyError, L.BdmHipError) as e:   # an entry point without a thunk / prototype: stay on the eager path
                self.tape.broken = f"finalize: {e!r}"
        return False


# ---- stream / event edges (the side streams of pvcnn.plan_sampling_chain and PVConv's point branch) -----------------------
def wait_stream(waiter, other):
    waiter.wait_stream(other)
    if _active is not None:
        _active.calls.append((_py, (None, waiter.wait_stream, (other,), {})))
        _active.native[len(_active.calls) - 1] = ("wait_stream", waiter, other)
        _active.keep.extend((waiter, other))


def record_event(event, stream):
    event.record(stream)
    if _active is not None:
        _active.calls.append((_py, (None, event.record, (stream,), {})))
        _active.native[len(_active.calls) - 1] = ("event_record", event, stream)
        _active.keep.extend((event, stream))


def wait_event(event):
    """The CURRENT stream waits for `event` (torch's `event.wait()`)."""
    s = torch.cuda.current_stream()
    s.wait_event(event)
    if _active is not None:
        _active.calls.append((_py, (None, s.wait_event, (event,), {})))
        _active.native[len(_active.calls) - 1] = ("event_wait", s, event)
        _active.keep.extend((s, event))
