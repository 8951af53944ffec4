import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.pvd import prepare_pvd_model
from bdm_amd.utils.procedural import fill_module_
from bdm_amd import sampling
B, N = 16, 4096
cfg = ProjectConfig(); cfg.dataset.max_points = N
cfg.aux_run.milestones, cfg.aux_run.roll_step = [1000, 968, 936, 872, 128, 64, 32, 0], 16
torch.manual_seed(cfg.run.seed)
model = fill_module_(get_model(cfg).eval(), seed=cfg.run.seed).cuda()
pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cuda")
batch = next(iter(SyntheticShapes(range(B), B, seed=cfg.run.seed, image_size=224, num_points=N))).to("cuda")
orig_is, orig_pp, orig_bl = model.interaction_sample, sampling.pvd_prior, sampling.blend_select
def stat(tag, x):
    torch.cuda.synchronize()
    print(tag, "finite", bool(torch.isfinite(x).all()), "absmax", float(x.abs().max()), flush=True)
    return x
model.interaction_sample = lambda *a, **k: stat(f"recon {k.get('start_time')}->{k.get('end_time')}", orig_is(*a, **k))
sampling.pvd_prior = lambda m, p, start_time, end_time: stat(f"prior {start_time}->{end_time}", orig_pp(m, p, start_time, end_time))
sampling.blend_select = lambda r, p, i: stat("blend", orig_bl(r, p, i))
gen = torch.Generator().manual_seed(cfg.run.seed)
out = sampling.bdm_blending(None, batch, cfg, model, pvd, generator=gen).points_padded()
stat("final", out)
