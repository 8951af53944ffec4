"""sha256 of one PC2 and one PVD forward on fixed inputs (B=2, N=2048): A/B check that a library change keeps the bits (BDM_LIB_PATH=...)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bdm_amd.pvcnn as PV
from bdm_amd.utils.procedural import fill_module_

B, N = int(os.environ.get("TB", 2)), int(os.environ.get("TN", 2048))
g = torch.Generator().manual_seed(3)
for which in ("pc2", "pvd"):
    net = fill_module_((PV.PVCNN2_PC2(3, 64, extra_feature_channels=32) if which == "pc2" else PV.PVCNN2_PVD(3, 64, extra_feature_channels=0)).eval(), seed=9).cuda()
    x = torch.cat([torch.randn(B, 3, N, generator=g) * 0.4, torch.randn(B, 32 if which == "pc2" else 0, N, generator=g)], dim=1).cuda()
    t = torch.tensor([900, 3][:B] if B <= 2 else [500] * B).cuda()
    out = net(x, t)
    print(which, B, N, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16])
