"""File output of the sample_* jobs.  The reference writes point clouds with pytorch3d.io.IO().save_pointcloud
(main_blending.py:427-445; binary little-endian PLY with float32 x, y, z -- pytorch3d's default) and the input image with
torchvision's to_pil_image(...).save(png) (main_blending.py:447-455).  Host-side, outside the timed path."""
import os

import numpy as np


def save_pointcloud_ply(points, path, binary=True):
    """float32 vertices, exact round trip in the (default) binary form."""
    pts = np.ascontiguousarray(np.asarray(points, dtype="<f4").reshape(-1, 3))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    fmt = "binary_little_endian" if binary else "ascii"
    header = (f"ply\nformat {fmt} 1.0\nelement vertex {pts.shape[0]}\nproperty float x\nproperty float y\n"
              "property float z\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        if binary:
            f.write(pts.tobytes())
        else:
            np.savetxt(f, pts, fmt="%.9g")


def load_pointcloud_ply(path):
    """(n, 3) float32 from an ASCII or binary-little-endian PLY whose vertex element starts with float x, y, z."""
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    head = raw[:end].decode("ascii").split("\n")
    n = int([l for l in head if l.startswith("element vertex")][0].split()[-1])
    fmt = [l for l in head if l.startswith("format")][0].split()[1]
    props = [l.split() for l in head if l.startswith("property")]
    if fmt == "ascii":
        rows = raw[end:].decode("ascii").split("\n")[:n]
        return np.array([[float(v) for v in r.split()[:3]] for r in rows], dtype=np.float32).reshape(-1, 3)
    if fmt != "binary_little_endian":
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    sizes = {"float": 4, "float32": 4, "double": 8, "float64": 8, "uchar": 1, "uint8": 1, "int": 4, "int32": 4}
    stride = sum(sizes[p[1]] for p in props)
    if [p[1] for p in props[:3]] not in (["float"] * 3, ["float32"] * 3):
        raise ValueError(f"{path}: vertex element must start with float x, y, z")
    body = np.frombuffer(raw, dtype=np.uint8, count=n * stride, offset=end).reshape(n, stride)
    return np.ascontiguousarray(body[:, :12]).view("<f4").reshape(n, 3).astype(np.float32)


def save_image_png(image_chw, path):
    """torchvision.transforms.functional.to_pil_image(float CHW in [0, 1]).save(path): pic.mul(255).byte(), mode RGB."""
    from PIL import Image
    arr = np.asarray(image_chw, dtype=np.float32)
    arr = (arr * 255.0).astype(np.uint8).transpose(1, 2, 0)  # truncation, as Tensor.byte() does
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr).save(path)
