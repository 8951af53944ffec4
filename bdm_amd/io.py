"""Point-cloud file output for the sample_* jobs (the reference uses pytorch3d.io.IO.save_pointcloud,
main_blending.py:427-445): ASCII PLY with float vertices.  Host-side, outside the timed path."""
import os

import numpy as np


def save_pointcloud_ply(points, path):
    pts = np.asarray(points, dtype=np.float32).reshape(-1, 3)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"element vertex {pts.shape[0]}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        np.savetxt(f, pts, fmt="%.7g")


def load_pointcloud_ply(path):
    with open(path) as f:
        lines = f.read().split("\n")
    n = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
    start = lines.index("end_header") + 1
    return np.loadtxt(lines[start:start + n], dtype=np.float32).reshape(-1, 3)
