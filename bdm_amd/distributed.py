"""One process per GPU; the sampling batch is sharded by shape, no collective inside a shape
(SURVEY.md 8e).  `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the
CPU tests of the host logic."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # BDM_DIST_BACKEND=gloo: used to exercise the multi-rank path on a single-GPU box
            backend = os.environ.get("BDM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if os.environ.get("BDM_SHARE_GPU") == "1":  # test aid: every rank drives cuda:0 (with BDM_DIST_BACKEND=gloo) on a 1-GPU box
        local_rank = 0
    return rank, local_rank, world


import contextlib


@contextlib.contextmanager
def gpu_turn(device=None):
    """Test aid for BDM_SHARE_GPU=1 (several ranks driving ONE GPU, used to exercise the multi-rank code on a 1-GPU box):
    the ranks take turns on the device (an flock'ed file; the GPU is drained before the turn ends).  Measured on MI355X /
    ROCm 7.2 (round 3): a kernel CO-RESIDENT with `sparse_gemm_s3_kernel` -- in another process or on another
    stream of the same process -- occasionally reads a wrong 64-byte sector (16 consecutive floats; all inputs bit-identical,
    the same launch alone is always right).  Round 4: the default forward launches that GEMM ONCE per forward (the 128 -> 64 first convolution of the 16^3 level; the 32^3 levels
    run one output-stationary kernel instead), and the side-stream kernels were run as victims next to both kernels
    (tools/coresidency/side_stream_victims.sh, profiles/r04_side_stream_victims.txt).  The lock stays the default of this TEST
    mode because two whole samplers sharing a GPU interleave every kernel pair, not only the pairs a forward produces.  No-op otherwise."""
    if os.environ.get("BDM_SHARE_GPU") != "1" or os.environ.get("BDM_GPU_TURN", "1") == "0":
        yield
        return
    import fcntl
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"bdm_share_gpu_{os.environ.get('MASTER_PORT', '0')}.lock")
    with open(path, "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
            if torch.cuda.is_available():
                torch.cuda.synchronize(device)
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def shard_indices(num_shapes, rank, world):
    """Contiguous, balanced shard of global shape indices [0, num_shapes) for `rank`."""
    base, extra = divmod(num_shapes, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def shared_run_dir(cfg, rank, world):
    """${run.save_dir}/${run.name}/<timestamp> chosen ONCE (rank 0) and shared, so that all ranks write into one tree."""
    from .config import run_dir
    box = [run_dir(cfg) if rank == 0 else None]
    if world > 1 and dist.is_initialized():
        dist.broadcast_object_list(box, src=0)
    return box[0]


def barrier():
    if dist.is_initialized():
        dist.barrier()


def gather_clouds(local, num_shapes, rank, world):
    """All ranks' (B_local, N, 3) results -> (num_shapes, N, 3) on every rank, in global shape order.
    The only collective on the sampling path (<= 6.3 MB per rank at B=256, N=16384)."""
    if world == 1 or not dist.is_initialized():
        return local
    counts = [len(shard_indices(num_shapes, r, world)) for r in range(world)]
    n, d = local.shape[1], local.shape[2]
    pad = max(counts)
    buf = torch.zeros(pad, n, d, dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def per_rank_values(value: float, device):
    """[value of rank 0, ..., value of rank world-1] on every rank (one all_gather of a double; bench.py's `per_rank_s`)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _amdgpu_devices_in_hip_order(sysfs):
    """PCI device directories of the amdgpu GPUs in the order HIP numbers them with no *_VISIBLE_DEVICES set: the KFD topology's
    GPU nodes (simd_count > 0) in node order, each mapped to its PCI device through its DRM render minor.  Falls back to the render
    nodes sorted by PCI address when the KFD topology is not readable (-> (devices, "kfd" | "pci"))."""
    import glob
    by_minor = {}
    for rd in glob.glob(os.path.join(sysfs, "class/drm/renderD*")):
        real = os.path.realpath(os.path.join(rd, "device"))
        try:
            vendor = open(os.path.join(real, "vendor")).read().strip()
            minor = int(os.path.basename(rd)[len("renderD"):])
        except (OSError, ValueError):
            continue
        if vendor == "0x1002":
            by_minor[minor] = real
    kfd = []
    nodes = glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*/properties"))
    for prop in sorted(nodes, key=lambda q: int(os.path.basename(os.path.dirname(q)))):
        try:
            kv = dict(line.split()[:2] for line in open(prop).read().splitlines() if len(line.split()) >= 2)
            if int(kv.get("simd_count", "0")) > 0:
                kfd.append(by_minor[int(kv["drm_render_minor"])])
        except (OSError, ValueError, KeyError):
            return sorted(set(by_minor.values()), key=os.path.basename), "pci"
    if kfd:
        return kfd, "kfd"
    return sorted(set(by_minor.values()), key=os.path.basename), "pci"


def _visible_ordinals(n_devices):
    """The device list left by ROCR_VISIBLE_DEVICES (applied first, by the runtime) and then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    (applied on top, by HIP), as indices into the unrestricted HIP order; None when a list is set but is not plain ordinals (UUIDs)."""
    order = list(range(n_devices))
    for names in (("ROCR_VISIBLE_DEVICES",), ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        text = next((os.environ[k] for k in names if os.environ.get(k, "") != ""), None)
        if text is None:
            continue
        picked = []
        for part in text.split(","):
            part = part.strip()
            if not part.isdigit():
                return None
            if int(part) >= len(order):
                break   # (the runtimes stop at the first invalid ordinal)
            picked.append(order[int(part)])
        order = picked
    return order


def pin_to_gpu_numa_node(local_rank, sysfs="/sys"):
    """Pin this worker to the host cores of the NUMA node its GPU hangs off (before any GPU call): the enqueue thread of a rank
    then never runs across the socket from its device.  HIP ordinal -> PCI device: the KFD topology's GPU nodes in node order
    (what HIP enumerates; PCI-address order only as a fallback when KFD is unreadable, reported as "order": "pci"), filtered by
    ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES when they hold plain ordinals -- a list that cannot be interpreted (UUIDs) means NO pinning
    rather than a guess.  The affinity is intersected with the mask the process already has (cgroup / taskset).  Returns a dict for
    the bench line ({"numa_node", "cpus", "pci", "order"}), or {"numa_node": None, "why": ...} -- never raises."""
    try:
        cards, order = _amdgpu_devices_in_hip_order(sysfs)
        visible = _visible_ordinals(len(cards))
        if visible is None:
            return {"numa_node": None, "why": "a *_VISIBLE_DEVICES list that is not plain ordinals: device order unknown, not pinning"}
        cards = [cards[i] for i in visible]
        if local_rank >= len(cards):
            return {"numa_node": None, "why": f"{len(cards)} visible amdgpu devices, local rank {local_rank}"}
        node = int(open(os.path.join(cards[local_rank], "numa_node")).read().strip())
        if node < 0:
            return {"numa_node": None, "why": "device reports no NUMA node"}
        cpus = _cpulist(open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read())
        allowed = os.sched_getaffinity(0)
        mine = sorted(cpus & allowed)
        if not mine:
            return {"numa_node": node, "why": "no allowed CPU on that node", "cpus": 0}
        os.sched_setaffinity(0, mine)
        return {"numa_node": node, "cpus": len(mine), "pci": os.path.basename(cards[local_rank]), "order": order}
    except (OSError, ValueError) as e:
        return {"numa_node": None, "why": repr(e)}


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
