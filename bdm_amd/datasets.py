"""ShapeNet-R2N2 and Pix3D readers + collate for the sampling entry points.

Host-side restatement of experiments/dataset/shapenet_r2n2.py:46-612 and experiments/dataset/pix3d.py:32-271: same
constructor arguments, on-disk layout, sample keys (the FrameData-like dict of shapenet_r2n2.py:508-536), normalisation and
camera construction, so that `dataset=shapenet_r2n2 | pix3d` works in main_blending.py / main_merging.py when the data are
on disk.  The datasets themselves are not available offline (/root/reference/.MISSING_LARGE_BLOBS); tests run on a tiny
synthetic fixture written in the datasets' layout (tests/test_datasets.py).  Dependencies of the reference that are absent
here are replaced by what they are used for: open3d / trimesh -> a vertex reader for .obj / .ply; pytorch3d
`sample_points_from_meshes` -> area-weighted surface sampling (only for Pix3D `processed=False`).

Layout read (ShapeNet-R2N2):
  <r2n2_dir>/<split_file>                      {"train"|"test": {synset_id: {object_id: ...}}}
  <r2n2_dir>/<pc_dict>                         {"train"|"test": {synset_id: {object_id: "train"|"val"|"test"}}}
  <root_dir>/<synset_id>/<subdir>/<object_id>.npy                       (15000, 3) ShapeNetCore.v2.PC15k points
  <r2n2_dir>/<views_rel_path>/<synset_id>/<object_id>/rendering/NN.png + rendering_metadata.txt
Layout read (Pix3D): <root>/<pc_dict> list of {"img", "model", "category", "rot_mat", "trans_mat", "focal_length",
  "img_size", "bbox"}; processed data under <root with 'pix3d' -> 'pix3d_processed'>.
"""
import json
import os
import random
from collections import OrderedDict

import numpy as np
import torch

from .cameras import PerspectiveCameras, pix3d_camera

R2N2_CATE = {"02691156": "airplane", "02828884": "bench", "02933112": "cabinet", "02958343": "car", "03001627": "chair",
             "03211117": "display", "03636649": "lamp", "03691459": "loudspeaker", "04090263": "rifle", "04256520": "sofa",
             "04379243": "table", "04401088": "telephone", "04530566": "watercraft"}
R2N2_SYNSET = {v: k for k, v in R2N2_CATE.items()}
R2N2_FOCAL = 2.1875  # K[0, 0] = K[1, 1] of shapenet_r2n2.py:46-53
MAX_CAMERA_DISTANCE = 1.75


def compute_extrinsic_matrix(azimuth, elevation, distance):
    """dataset/utils.py:40-86 (from meshrcnn): 4x4 world -> camera matrix of a camera looking at the origin."""
    import math
    az, el = -math.pi * float(azimuth) / 180.0, -math.pi * float(elevation) / 180.0
    sa, ca, se, ce = math.sin(az), math.cos(az), math.sin(el), math.cos(el)
    R_world2obj = torch.tensor([[ca * ce, sa * ce, -se], [-sa, ca, 0], [ca * se, sa * se, ce]])
    R_obj2cam = torch.tensor([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
    R_world2cam = R_obj2cam.mm(R_world2obj)
    T_world2cam = -(R_obj2cam.mm(torch.tensor([[float(distance), 0, 0]]).t()))
    RT = torch.cat([torch.cat([R_world2cam, T_world2cam], dim=1), torch.tensor([[0.0, 0, 0, 1]])])
    rot = torch.tensor([[1.0, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])  # Blender's .obj axis quirk (:81-84)
    return RT.mm(rot)


def compute_camera_calibration(RT):
    """dataset/utils.py:89-114: (R, T) of the ShapeNet -> PyTorch3D-world camera."""
    s2p = torch.tensor([[-1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, -1.0, 0], [0, 0, 0, 1.0]])
    RT = torch.transpose(RT, 0, 1).mm(s2p)
    return RT[:3, :3], RT[3, :3]


def transform_v2_to_v1(pc):
    """shapenet_r2n2.py:56-62."""
    out = pc.clone()
    out[:, 0], out[:, 1], out[:, 2] = -pc[:, 2], pc[:, 1], -pc[:, 0]
    return out.float()


def build_camera_from_R2N2(Rs, Ts, mean, std):
    """shapenet_r2n2.py:65-95: folds the dataset normalisation (x - mean) / std into the camera."""
    pose = torch.cat([Rs, Ts[None]], dim=0)
    extrin = torch.cat([pose, torch.tensor([[0.0, 0, 0, 1]]).T], dim=1)
    s2p = torch.tensor([[-1.0, 0, 0, 0], [0, -1.0, 0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]])
    RT = extrin @ s2p
    R = RT[:3, :3].clone()
    camera_R = (R * std).float()
    camera_T = (mean @ R / std + RT[3, :3].clone()).float()
    camera_R[:, :2] *= -1
    camera_T[:2] *= -1
    return PerspectiveCameras(focal_length=torch.tensor([[R2N2_FOCAL, R2N2_FOCAL]]), principal_point=torch.tensor([[0.0, 0.0]]),
                              R=camera_R[None], T=camera_T[None])


def _load_image(path, size):
    """load_data's image branch (shapenet_r2n2.py:372-384): drop alpha, bilinear resize, [0, 1] float CHW."""
    from PIL import Image
    raw = Image.open(path)
    bands = raw.split()
    img = Image.merge("RGB", bands[:3]).resize((size, size), Image.BILINEAR)
    return torch.from_numpy(np.array(img) / 255.0)[..., :3].permute(2, 0, 1).float()


def _frame(**kw):
    s = OrderedDict()
    for k in ("frame_number", "sequence_name", "sequence_category", "frame_timestamp", "image_size_hw",
              "effective_image_size_hw", "image_path", "image_rgb", "mask_crop", "depth_path", "depth_map", "depth_mask",
              "mask_path", "fg_probability", "bbox_xywh", "crop_bbox_xywh", "camera", "camera_quality_score",
              "point_cloud_quality_score", "sequence_point_cloud_path", "sequence_point_cloud", "sequence_point_cloud_idx",
              "frame_type", "meta"):
        s[k] = kw.get(k)
    return s


class ShapeNet_R2N2(torch.utils.data.Dataset):
    """shapenet_r2n2.py:98-536 (build_data; the multiprocessing variant builds the same lists)."""

    def __init__(self, root_dir, r2n2_dir, pc_dict="pc_dict_v2.json", split_file="R2N2_split.json",
                 views_rel_path="ShapeNetRendering", which_view_from24=("00",), categories=("chair",), split="train",
                 sample_size=4096, img_size=224, scale_factor=1.0, random_subsample=True, normalize_per_shape=False,
                 box_per_shape=False, subset_ratio=1.0, start_ratio=0.0, input_dim=3):
        if split not in ("train", "test"):
            raise ValueError("split has to be one of (train, test).")
        self.root_dir, self.r2n2_dir, self.views_rel_path, self.split = root_dir, r2n2_dir, views_rel_path, split
        self.cate_id = list(R2N2_SYNSET.values()) if "all" in categories else [R2N2_SYNSET[c] for c in categories]
        with open(os.path.join(r2n2_dir, split_file)) as f:
            self.split_dict = json.load(f)
        with open(os.path.join(r2n2_dir, pc_dict)) as f:
            self.pc_subdir = json.load(f)
        if not os.path.isdir(os.path.join(r2n2_dir, views_rel_path)):
            raise FileNotFoundError(os.path.join(r2n2_dir, views_rel_path))
        self.img_size, self.scale_factor, self.sample_size = img_size, scale_factor, sample_size
        self.which_view_from24 = list(which_view_from24)
        self.normalize_per_shape, self.random_subsample, self.input_dim = normalize_per_shape, random_subsample, input_dim
        self.subset_ratio, self.start_ratio = subset_ratio, start_ratio
        self.build_data()

    def load_data(self, point_clouds_path, rendering_path, metadata_lines, view_number):
        image_path = os.path.join(rendering_path, view_number + ".png")
        image = _load_image(image_path, self.img_size)
        pc_v2 = torch.tensor(np.load(point_clouds_path))
        if pc_v2.shape[0] != 15000:
            raise FileNotFoundError(f"{point_clouds_path}: expected (15000, 3) points")
        azim, elev, _yaw, dist_ratio, _fov = [float(v) for v in metadata_lines[int(view_number)].strip().split(" ")]
        Rs, Ts = compute_camera_calibration(compute_extrinsic_matrix(azim, elev, dist_ratio * MAX_CAMERA_DISTANCE))
        return image_path, image, point_clouds_path, transform_v2_to_v1(pc_v2), Rs, Ts

    def build_data(self):
        self.camera, self.img_rgb, self.img_path, self.point_clouds_path, self.Rs, self.Ts = [], [], [], [], [], []
        clouds = []
        for cate_id in self.cate_id:
            object_ids = list(self.split_dict[self.split][cate_id].keys())
            object_ids = object_ids[: int(len(object_ids) * self.subset_ratio)]
            for object_id in object_ids:
                if object_id not in self.pc_subdir[self.split][cate_id]:
                    continue  # in the R2N2 split but not in ShapeNetCore.v2.PC15k
                pc_path = os.path.join(self.root_dir, cate_id, self.pc_subdir[self.split][cate_id][object_id], object_id + ".npy")
                rendering_path = os.path.join(self.r2n2_dir, self.views_rel_path, cate_id, object_id, "rendering")
                with open(os.path.join(rendering_path, "rendering_metadata.txt")) as f:
                    metadata_lines = f.readlines()
                for view in self.which_view_from24:
                    img_path, img, pcp, pc, Rs, Ts = self.load_data(pc_path, rendering_path, metadata_lines, view)
                    self.img_path.append(img_path); self.img_rgb.append(img); self.point_clouds_path.append(pcp)
                    clouds.append(pc); self.Rs.append(Rs); self.Ts.append(Ts)
        order = list(range(len(clouds)))
        random.Random(38383).shuffle(order)  # shapenet_r2n2.py:446-447
        clouds = [clouds[i][None] for i in order]
        for name in ("point_clouds_path", "img_rgb", "img_path", "Rs", "Ts"):
            setattr(self, name, [getattr(self, name)[i] for i in order])
        self.all_points = torch.cat(clouds, dim=0) if clouds else torch.zeros(0, 15000, 3)
        if self.normalize_per_shape:
            B = self.all_points.shape[0]
            self.all_points_mean = self.all_points.mean(axis=1).reshape(B, 1, self.input_dim)
            self.all_points_std = self.all_points.reshape(B, -1).std(axis=1).reshape(B, 1, 1)
        else:  # across the whole dataset (:467-476)
            self.all_points_mean = self.all_points.reshape(-1, self.input_dim).mean(axis=0).reshape(1, 1, self.input_dim)
            self.all_points_std = self.all_points.reshape(-1).std(axis=0).reshape(1, 1, 1)
        self.all_points = (self.all_points - self.all_points_mean) / self.all_points_std
        self.all_point_clouds = []
        for i in range(self.all_points.shape[0]):
            pc = self.all_points[i]
            if self.random_subsample:  # numpy's global generator, seeded by the entry point (training_utils.py:373-386)
                pc = pc[np.random.choice(pc.shape[0], self.sample_size), :].float()
            self.all_point_clouds.append(pc)
            k = i if self.normalize_per_shape else 0
            self.camera.append(build_camera_from_R2N2(self.Rs[i].clone(), self.Ts[i].clone(), self.all_points_mean[k, 0, :],
                                                      self.all_points_std[k, 0, :]))

    def __len__(self):
        return len(self.img_path)

    def __getitem__(self, idx):
        p = self.img_path[idx].split("/")
        frame = p[-1].split(".")[0]
        hw = torch.tensor(self.img_rgb[idx].shape[1:]).long()
        return _frame(frame_number=frame, sequence_name=p[-3] + "_" + frame, sequence_category=R2N2_CATE[p[-4]],
                      frame_timestamp=0, image_size_hw=hw, effective_image_size_hw=hw, image_path=self.img_path[idx],
                      image_rgb=self.img_rgb[idx], camera=self.camera[idx],
                      sequence_point_cloud_path=self.point_clouds_path[idx], sequence_point_cloud=self.all_point_clouds[idx],
                      sequence_point_cloud_idx=0, frame_type="real", meta={"dataset_index": idx})


# ---------------------------------------------------------------------------------------------------------------
def read_vertices(path):
    """Vertices of a .obj / .ply / .npy file (what the reference reads through trimesh.load(...).vertices)."""
    if path.endswith(".npy"):
        return np.load(path).astype(np.float64).reshape(-1, 3)
    if path.endswith(".ply"):
        from .io import load_pointcloud_ply
        return load_pointcloud_ply(path).astype(np.float64)
    verts = []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(v) for v in line.split()[1:4]])
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3)


def sample_points_from_obj(path, n, generator=None):
    """Area-weighted surface samples of a triangle mesh (stands in for pytorch3d.ops.sample_points_from_meshes, pix3d.py:84)."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(v) for v in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) - 1 for tok in line.split()[1:]]
                faces.extend([[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)])  # fan triangulation
    v, f = torch.tensor(verts, dtype=torch.float64), torch.tensor(faces, dtype=torch.long)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = torch.linalg.cross(b - a, c - a).norm(dim=1) * 0.5
    pick = torch.multinomial(area / area.sum(), n, replacement=True, generator=generator)
    u = torch.rand(n, 2, dtype=torch.float64, generator=generator)
    su = u[:, 0].sqrt()
    w0, w1, w2 = 1 - su, su * (1 - u[:, 1]), su * u[:, 1]
    return (w0[:, None] * a[pick] + w1[:, None] * b[pick] + w2[:, None] * c[pick]).numpy()


class Pix3D(torch.utils.data.Dataset):
    """pix3d.py:32-233."""

    def __init__(self, root_dir, split="train", sample_size=4096, img_size=224, pc_dict="pix3d.json", category="chair",
                 subset_ratio=1.0, processed=True):
        with open(os.path.join(root_dir, pc_dict)) as f:
            cat_json = [x for x in json.load(f) if x["category"] == category]
        cut = int(len(cat_json) * 0.8)  # 4:1 split in file order (:50-60)
        if split == "train":
            self.data = cat_json[:cut]
            if subset_ratio != 1.0:
                self.data = self.data[: int(len(self.data) * subset_ratio)]
        elif split == "test":
            self.data = cat_json[cut:]
        else:
            raise ValueError("split must be 'train' or 'test'")
        self.root_dir, self.processed = root_dir, processed
        # pix3d.py:66 uses str.replace("pix3d", "pix3d_processed") on the whole path; only the LAST occurrence is replaced
        # here (identical for the recipe's .../Pix3D/pix3d, and not confused by a parent directory that contains "pix3d")
        head, sep, tail = root_dir.rpartition("pix3d")
        self.processed_root_dir = head + "pix3d_processed" + tail if sep else root_dir
        self.sample_size, self.img_size = sample_size, img_size

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        from PIL import Image
        s = self.data[idx]
        if self.processed:
            pts = read_vertices(os.path.join(self.processed_root_dir, s["model"]))
        else:
            pts = sample_points_from_obj(os.path.join(self.root_dir, s["model"]), self.sample_size)
        m = pts.mean(axis=0)
        std = pts.reshape(1, -1).std(axis=1)
        pts_norm = (pts - m) / std
        v2_to_v1 = np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0]])
        pts_v1 = (v2_to_v1 @ pts_norm.T).T
        camera = pix3d_camera(s["rot_mat"], s["trans_mat"], m, float(std[0]), s["img_size"], s["bbox"], s["focal_length"],
                              out_size=self.img_size)
        if self.processed:
            img = Image.open(os.path.join(self.processed_root_dir, s["img"]))
        else:
            x0, y0, x1, y1 = s["bbox"]
            cx, cy, half = (x0 + x1) / 2, (y0 + y1) / 2, max(y1 - y0, x1 - x0) / 2
            img = Image.open(os.path.join(self.root_dir, s["img"])).crop((cx - half, cy - half, cx + half, cy + half)) \
                .resize((self.img_size, self.img_size))
        if img.mode != "RGB":
            img = img.convert("RGB")
        image = torch.from_numpy(np.array(img) / 255.0)[..., :3].permute(2, 0, 1).float()
        frame = s["img"].split("/")[-1].split(".")[0]
        base = self.processed_root_dir if self.processed else self.root_dir
        return _frame(frame_number=frame, sequence_name=s["model"].split("/")[-2] + "_" + frame, sequence_category=s["category"],
                      frame_timestamp=0, image_size_hw=torch.tensor([s["img_size"][1], s["img_size"][0]]).long(),
                      effective_image_size_hw=torch.tensor([self.img_size, self.img_size]).long(),
                      image_path=os.path.join(base, s["img"]), image_rgb=image, camera=camera,
                      sequence_point_cloud_path=os.path.join(base, s["model"]), sequence_point_cloud=torch.tensor(pts_v1).float(),
                      sequence_point_cloud_idx=0, frame_type="real", meta={"dataset_index": idx})


# ---------------------------------------------------------------------------------------------------------------
def custom_collate(batch):
    """shapenet_r2n2.py:601-612 / pix3d.py:273-285: cameras stay a python LIST, None stays None, the rest is stacked."""
    data = {}
    for key in batch[0].keys():
        if isinstance(batch[0][key], PerspectiveCameras):
            data[key] = [sample[key] for sample in batch]
        elif batch[0][key] is None:
            data[key] = None
        else:
            data[key] = torch.utils.data.dataloader.default_collate([sample[key] for sample in batch])
    return data


def accelerate_batch_shard(num_samples, batch_size, rank, world):
    """The index batches rank `rank` of `world` processes sees when accelerate shards a sequential, drop_last=False
    DataLoader the way `accelerator.prepare(dataloader)` does by default in the reference (main_blending.py:115-124;
    accelerate.data_loader.BatchSamplerShard, split_batches=False, even_batches=True): whole batches dealt round-robin
    (batch i -> rank i % world); when the batch count is not a multiple of `world`, or the last batch is short, the tail is
    completed with samples from the START of the dataset so that every rank gets the same number of full batches.
    Restated here (accelerate is not a dependency of the sampling path); tests/test_datasets.py checks it against the
    installed accelerate where that is importable."""
    batches = [list(range(i, min(i + batch_size, num_samples))) for i in range(0, num_samples, batch_size)]
    if world == 1:
        return batches
    mine, initial, pending, idx, batch = [], [], None, -1, []
    for idx, batch in enumerate(batches):
        if idx < world:
            initial += batch
        if idx % world == rank:
            pending = batch
        if idx % world == world - 1 and len(batch) == batch_size:
            mine.append(pending)
            pending = None
    if initial:
        if pending and len(pending) == batch_size:
            mine.append(pending)
        while len(initial) < world * batch_size:
            initial += initial
        if len(batch) == batch_size:
            batch, idx = [], idx + 1
        batch, cycle = list(batch), 0
        while idx % world != 0 or len(batch) > 0:
            end = cycle + batch_size - len(batch)
            batch += initial[cycle:end]
            if idx % world == rank:
                mine.append(batch)
            cycle, batch, idx = end, [], idx + 1
    return mine


def get_dataset(cfg, rank=0, world=1):
    """dataset/__init__.py:get_dataset for the sample_* jobs: (None, dataloader_val, dataloader_vis).  With world > 1 the
    validation loader is sharded as accelerator.prepare(dataloader) shards it in the reference (main_blending.py:115-124):
    whole batches round-robin over the ranks, tail padded from the start of the dataset (`accelerate_batch_shard`), so the
    shape -> rank assignment of a multi-GPU run is the reference's.  meta['dataset_index'] stays the GLOBAL index (the key of the
    per-shape random streams).  As in the reference every rank builds the full dataset with numpy seeded `seed + rank`
    (training_utils.py:373-378), so the random subsample of a ground-truth cloud depends on the rank that loads it."""
    d, dl = cfg.dataset, cfg.dataloader
    if d.type == "shapenet_r2n2":
        ds = ShapeNet_R2N2(root_dir=d.root, r2n2_dir=d.r2n2_dir, pc_dict=d.pc_dict or "pc_dict_v2.json", split_file=d.split_file,
                           views_rel_path=d.views_rel_path, which_view_from24=[d.which_view_from24], categories=[d.category],
                           sample_size=d.max_points, split="test", img_size=d.image_size, scale_factor=d.scale_factor,
                           random_subsample=True)
    elif d.type == "pix3d":
        ds = Pix3D(root_dir=d.root, pc_dict=d.pc_dict or "pix3d.json", category=d.category,
                   split="test", sample_size=d.max_points, img_size=d.image_size, processed=d.processed)
    else:
        raise NotImplementedError(d.type)
    if world > 1:
        val = torch.utils.data.DataLoader(ds, batch_sampler=accelerate_batch_shard(len(ds), dl.batch_size, rank, world), num_workers=0,
                                          collate_fn=custom_collate)
    else:
        val = torch.utils.data.DataLoader(ds, batch_size=dl.batch_size, shuffle=False, num_workers=0, drop_last=False,
                                          collate_fn=custom_collate)
    return None, val, val
