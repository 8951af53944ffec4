"""Where a C2 trajectory's time goes besides its replayed steps: wall time (device-synchronised) of every PC2 loop segment, every prior segment and
of everything else (ViT of the conditioning image, centring / blending, host set-up).  python tools/trajectory_breakdown.py [B=16] [N=4096]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bdm_amd import sampling
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.model import get_model
from bdm_amd.pvd import prepare_pvd_model
from bdm_amd.sampling import batch_streams, bdm_blending
from bdm_amd.utils.procedural import fill_module_

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda")
cfg = ProjectConfig(); cfg.dataset.max_points = N; cfg.run.rng = "per_shape"
model = fill_module_(get_model(cfg).eval(), seed=cfg.run.seed).to(dev)
pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, dev)
batch = next(iter(SyntheticShapes(range(B), B, seed=cfg.run.seed, image_size=224, num_points=N))).to(dev)
acc = {"pc2": [0.0, 0, 0], "pvd": [0.0, 0, 0]}

def timed(fn, key, steps_of):
    def wrapper(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize(); acc[key][0] += time.perf_counter() - t0; acc[key][1] += 1; acc[key][2] += steps_of(*a, **k)
        return out
    return wrapper

model._denoise_loop = timed(model._denoise_loop, "pc2", lambda x, cam, img, m, sch, ts, generator=None: len(ts))
sampling.pvd_prior = timed(sampling.pvd_prior, "pvd", lambda m, p, start_time, end_time: start_time - end_time)
for rep in range(3):
    for v in acc.values(): v[0] = 0.0; v[1] = 0; v[2] = 0
    model._cond_cache = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bdm_blending(None, batch, cfg, model, pvd, streams=batch_streams(cfg, batch, dev, sample_idx=rep + 1)).points_padded()
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    rest = total - acc["pc2"][0] - acc["pvd"][0]
    print(f"trajectory {rep}: {total * 1e3:8.1f} ms | PC2 loops {acc['pc2'][0] * 1e3:8.1f} ms in {acc['pc2'][1]} segments, {acc['pc2'][2]} steps "
          f"({acc['pc2'][0] / max(acc['pc2'][2], 1) * 1e3:.3f} ms / step) | prior {acc['pvd'][0] * 1e3:7.1f} ms in {acc['pvd'][1]} segments, {acc['pvd'][2]} steps "
          f"({acc['pvd'][0] / max(acc['pvd'][2], 1) * 1e3:.3f} ms / step) | everything else {rest * 1e3:6.1f} ms")
