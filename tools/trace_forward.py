"""One PC2 forward (B=16, N=4096) between two marker kernels, for `rocprofv3 --kernel-trace`: tools/trace_summary.py then
lists every launch of that forward in order with its duration and grid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.pvd import prepare_pvd_model
from bdm_amd.utils.procedural import fill_module_

B, N = int(os.environ.get("TB", 16)), int(os.environ.get("TN", 4096))
which = sys.argv[1] if len(sys.argv) > 1 else "pc2"
cfg = ProjectConfig(); cfg.dataset.max_points = N
model = fill_module_(get_model(cfg).eval(), seed=1).cuda()
pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cuda")
batch = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda") * float(os.environ.get("SCALE", "0.5"))
t = torch.full((B,), 500, dtype=torch.int64, device="cuda")
xin = model.get_input_with_conditioning(x, batch.camera, batch.image_rgb, None, t)
xp = x.transpose(1, 2).contiguous()
fn = (lambda: model.point_cloud_model(xin, t)) if which == "pc2" else (lambda: pvd.model(xp, t))
for _ in range(2): fn()
torch.cuda.synchronize()
marker = torch.zeros(7, device="cuda")
abi_log = os.environ.get("BDM_ABI_LOG")   # path: the ordered (function, integer arguments) list of the marked forward (tools/pmc_to_json.py)
calls = []
if abi_log:
    from bdm_amd import _lib as L

    class _Log:
        def __init__(self, h):
            self._h = h

        def __getattr__(self, name):
            f = getattr(self._h, name)
            if not name.startswith("bdm_") or name.endswith(("_bytes", "_elems", "_slices")) or name == "bdm_last_error":
                return f

            def call(*a):
                from bdm_amd import profiling as P
                spec = P.SPEC.get(name)
                try:     # the shape signature bench.py's roofline rows are keyed by (profiling.SPEC)
                    ints = [int(P._plain(v)) for v in spec[1](a)] if spec else []
                except (TypeError, ValueError):
                    ints = []
                calls.append((name, ints))
                return f(*a)
            return call
    saved = L.lib()
    L._lib = _Log(saved)
torch.cumsum(marker, 0); torch.cuda.synchronize()   # marker kernel 1
fn(); torch.cuda.synchronize()
torch.cumsum(marker, 0); torch.cuda.synchronize()   # marker kernel 2
if abi_log:
    import json
    L._lib = saved
    json.dump(calls, open(abi_log, "w"))
