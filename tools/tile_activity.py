"""How much of a voxel grid would an ACTIVE-TILE convolution have to touch?  

The voxeliser normalises a cloud so that its farthest point sits on the grid boundary (modules/voxelization.py:16-25), so the cloud
spans the whole grid however few voxels it occupies.  For clouds of the sampler's kind (x_t is Gaussian noise pulled towards a shape;
the bench's final clouds occupy 5.2 % of the 32^3 voxels, a unit Gaussian 6.3 %) this counts, per tile shape, the fraction of tiles that
hold at least one voxel of the once- (first convolution's output support) and twice-dilated (second convolution's) occupied set.
CPU only (numpy).   usage: tile_activity.py [n_points] [r] [trials]"""
import sys
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
r = int(sys.argv[2]) if len(sys.argv) > 2 else 32
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 8


def dilate(a):
    o = a.copy()
    for ax in range(3):
        lo, hi = o.copy(), o.copy()
        sl_dst, sl_src = [slice(None)] * 3, [slice(None)] * 3
        sl_dst[ax], sl_src[ax] = slice(1, None), slice(None, -1)
        lo[tuple(sl_dst)] |= o[tuple(sl_src)]
        hi[tuple(sl_src)] |= o[tuple(sl_dst)]
        o = lo | hi
    return o


def grid_of(pts):
    c = pts - pts.mean(0)
    v = (c / (2 * np.linalg.norm(c, axis=1).max()) + 0.5) * r
    v = np.clip(np.round(v), 0, r - 1).astype(int)
    occ = np.zeros((r, r, r), bool)
    occ[v[:, 0], v[:, 1], v[:, 2]] = True
    return occ


tiles = {"2x8x32 (conv3d_h2 at 32^3)": (2, 8, 32), "8x8x8": (8, 8, 8), "4x4x8": (4, 4, 8), "4x4x4": (4, 4, 4), "2x2x32": (2, 2, 32)}
tiles = {k: t for k, t in tiles.items() if all(r % d == 0 for d in t) and all(d <= r for d in t)}
rng = np.random.default_rng(0)
for label, draw in (("unit Gaussian", lambda: rng.standard_normal((n, 3))),
                    ("heavier-tailed (Gaussian x |Gaussian|)", lambda: rng.standard_normal((n, 3)) * np.abs(rng.standard_normal((n, 1))))):
    acc = {}
    for _ in range(trials):
        occ = grid_of(draw())
        d1 = dilate(occ)
        d2 = dilate(d1)
        for k, a in (("voxels occupied", occ), ("voxels once-dilated", d1), ("voxels twice-dilated", d2)):
            acc.setdefault(k, []).append(a.mean())
        for name, (tx, ty, tz) in tiles.items():
            for lab, a in (("once", d1), ("twice", d2)):
                t = a.reshape(r // tx, tx, r // ty, ty, r // tz, tz).any(axis=(1, 3, 5))
                acc.setdefault(f"tiles {name} touched by the {lab}-dilated set", []).append(t.mean())
    print(f"{label}: {n} points, {r}^3 grid, mean of {trials} clouds")
    for k, v in acc.items():
        print(f"   {k:64s} {100 * np.mean(v):5.1f} %")
