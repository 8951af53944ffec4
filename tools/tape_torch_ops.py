"""Which torch operators does a recorded reverse step still contain (each is 1 - 2 framework launches and a break in the native run)?
Prints them with argument shapes / strides and the Python frame that issued them.  python tools/tape_torch_ops.py [B] [N]"""
import os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import tape as T
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.utils.procedural import fill_module_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
seen = []
orig = T._TorchOps.__torch_dispatch__
def spy(self, func, types, args=(), kwargs=None):
    out = orig(self, func, types, args, kwargs)
    name = func._schema.name
    if not (func.is_view or name in T._ALLOC_ONLY):
        ts = [a for a in args if isinstance(a, torch.Tensor)]
        if any(t.is_cuda for t in ts) or (isinstance(out, torch.Tensor) and out.is_cuda):
            fr = [f for f in traceback.extract_stack() if "/bdm_amd/" in f.filename and "tape.py" not in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])
            seen.append((name, [(tuple(t.shape), tuple(t.stride()), str(t.dtype).replace("torch.", "")) for t in ts], where))
    return out
T._TorchOps.__torch_dispatch__ = spy
cfg = ProjectConfig(); cfg.dataset.max_points = N
model = fill_module_(get_model(cfg).eval(), seed=1).cuda()
b = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda") * 0.5
sched = model.schedulers_map["ddpm"]; sched.set_timesteps(1000)
model._denoise_loop(x, b.camera, b.image_rgb, None, sched, list(range(999, 987, -1)))
torch.cuda.synchronize()
tp = model._tape_cache["tape"]
print(f"tape: {len(tp)} entries, {len(tp.torch_ops)} torch operators, python entries after finalize: {getattr(tp, 'python_entries', '?')}")
for name, shapes, where in seen:
    print(f"  {name:<28s} {shapes}  @ {where}")
