import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.utils.procedural import fill_module_
from bdm_amd.sampling import _initial_cloud
B, N = 16, 4096
cfg = ProjectConfig(); cfg.dataset.max_points = N
torch.manual_seed(cfg.run.seed)
model = fill_module_(get_model(cfg).eval(), seed=cfg.run.seed).cuda()
batch = next(iter(SyntheticShapes(range(B), B, seed=cfg.run.seed, image_size=224, num_points=N))).to("cuda")
x = _initial_cloud(B, N, torch.device("cuda"))
sched = model.schedulers_map["ddpm"]; sched.set_timesteps(1000)
start = int(sys.argv[1]) if len(sys.argv) > 1 else 999
for t in range(start, -1, -1):
    x = model._denoise_loop(x, batch.camera, batch.image_rgb, None, sched, [t])
    torch.cuda.synchronize()
    if t % 20 == 0 or not bool(torch.isfinite(x).all()):
        a = x.abs()
        print(t, "finite", bool(torch.isfinite(x).all()), "absmax", float(a.max()), "mean", float(a.mean()), flush=True)
        if not bool(torch.isfinite(x).all()):
            break
