"""cProfile of the host side of the PC^2 reverse loop (no synchronisation inside): where the enqueue time goes."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bdm_amd.model as M
from bdm_amd.config import ProjectConfig
from bdm_amd.data import SyntheticShapes
from bdm_amd.utils.procedural import fill_module_
B, N, steps = int(os.environ.get("PB", 16)), int(os.environ.get("PN", 4096)), 30
cfg = ProjectConfig(); cfg.dataset.max_points = N
model = fill_module_(M.get_model(cfg).eval(), seed=1).cuda()
batch = next(iter(SyntheticShapes(range(B), B, num_points=N))).to("cuda")
x = torch.randn(B, N, 3, device="cuda")
model.interaction_sample(x.clone(), batch.camera, batch.image_rgb, None, start_time=900, end_time=895)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
model.interaction_sample(x.clone(), batch.camera, batch.image_rgb, None, start_time=800, end_time=800 - steps)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
