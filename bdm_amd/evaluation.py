"""Quality-parity harness: the reference's acceptance metrics on `sample_*/{gt,pred}` directories
(experiments/evaluation/evaluation_cd.py:111-131: pytorch3d chamfer_distance x 1000 on mean-centred clouds;
experiments/evaluation/evaluation_f1.py:90-110: F1 with threshold 0.01 on SQUARED nearest-neighbour distances).
Nearest-neighbour search runs in HIP (bdm_nn_sqdist); the scalar reductions are host-side numpy."""
import os

import numpy as np
import torch

from . import _lib as L
from .io import load_pointcloud_ply


def nn_sqdist(src, tgt):
    """src (B,N,3), tgt (B,M,3) on the GPU -> (B,N) squared distance to the nearest target point."""
    src, tgt = src.float().contiguous(), tgt.float().contiguous()
    B, N, _ = src.shape
    out = torch.empty(B, N, dtype=torch.float32, device=src.device)
    L.check(L.lib().bdm_nn_sqdist(B, N, tgt.shape[1], L.ptr(src), L.ptr(tgt), L.ptr(out), L.stream()), "nn_sqdist")
    return out


def chamfer_distance_x1000(pred, gt):
    """evaluation_cd.py: both clouds mean-centred; mean squared NN distance in both directions, summed, x 1000."""
    pred = pred - pred.mean(dim=1, keepdim=True)
    gt = gt - gt.mean(dim=1, keepdim=True)
    d_pg = nn_sqdist(pred, gt).double().cpu().numpy().mean(axis=1)
    d_gp = nn_sqdist(gt, pred).double().cpu().numpy().mean(axis=1)
    return (d_pg + d_gp) * 1000.0


def f1_score(pred, gt, thr=0.01):
    """evaluation_f1.py:90-110 (threshold on the squared distance, clamped at 1e-12, as the reference does)."""
    d1 = np.maximum(nn_sqdist(gt, pred).cpu().numpy(), 1e-12)
    d2 = np.maximum(nn_sqdist(pred, gt).cpu().numpy(), 1e-12)
    precision = (d1 < thr).mean(axis=1)
    recall = (d2 < thr).mean(axis=1)
    return 2 * recall * precision / (recall + precision + 1e-12)


def evaluate_dirs(pred_dir, gt_dir, device="cuda"):
    """Mean CD x 1e3 and mean F1@0.01 over matching .ply files of two directory trees."""
    cds, f1s = [], []
    for root, _, files in os.walk(pred_dir):
        for f in sorted(files):
            if not f.endswith(".ply"):
                continue
            rel = os.path.relpath(os.path.join(root, f), pred_dir)
            gt_path = os.path.join(gt_dir, rel)
            if not os.path.exists(gt_path):
                continue
            p = torch.from_numpy(load_pointcloud_ply(os.path.join(root, f)))[None].to(device)
            g = torch.from_numpy(load_pointcloud_ply(gt_path))[None].to(device)
            cds.append(float(chamfer_distance_x1000(p, g)[0]))
            pc, gc = p - p.mean(1, keepdim=True), g - g.mean(1, keepdim=True)
            f1s.append(float(f1_score(pc, gc)[0]))
    return {"num": len(cds), "cd_x1000": float(np.mean(cds)) if cds else float("nan"),
            "f1_at_0.01": float(np.mean(f1s)) if f1s else float("nan")}
