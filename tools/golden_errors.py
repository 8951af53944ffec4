"""Prints the relative L2 error of the HIP denoisers against the reference goldens for each arithmetic mode.
usage: BDM_CONV=fp16x3|bf16x6|fp32 python tools/golden_errors.py"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import point_cloud_inputs, rel_l2
from bdm_amd.pvcnn import PVCNN2_PC2, PVCNN2_PVD
from bdm_amd.utils.procedural import fill_module_
G = os.path.join(R, "tests", "golden")
for name, cls, kw, shape in [("pc2_full_n1024.npz", PVCNN2_PC2, dict(extra_feature_channels=387), (1, 390, 1024)),
                             ("pvd_full_n1024.npz", PVCNN2_PVD, dict(extra_feature_channels=0), (2, 3, 1024))]:
    g = np.load(os.path.join(G, name))
    net = fill_module_(cls(num_classes=3, embed_dim=64, **kw).eval(), seed=int(g["weight_seed"])).cuda()
    x = point_cloud_inputs(*shape, int(g["input_seed"]))
    y = net(x.cuda(), torch.from_numpy(g["t"]).cuda()).cpu()
    print(f"BDM_CONV={os.environ.get('BDM_CONV', 'fp16x3'):7s} {name:22s} rel L2 vs reference golden = {rel_l2(y, torch.from_numpy(g['out'])):.3e}")
