"""ShapeNet-R2N2 / Pix3D readers (bdm_amd/datasets.py; reference: experiments/dataset/shapenet_r2n2.py, pix3d.py) on a tiny
synthetic fixture written in the datasets' on-disk layout; the CLI end to end on that fixture (-m gpu)."""
import json
import os

import numpy as np
import pytest
import torch

SYN = "03001627"  # chair


def make_r2n2_fixture(base, n_test=3, n_points=15000):
    from PIL import Image
    root, r2n2 = base / "ShapeNetCore.v2.PC15k", base / "ShapeNet.R2N2"
    rng = np.random.default_rng(0)
    ids = {"train": ["tr%04d" % i for i in range(2)], "test": ["te%04d" % i for i in range(n_test)]}
    split = {s: {SYN: {i: list(range(24)) for i in v}} for s, v in ids.items()}
    sub = {"train": {SYN: {i: "train" for i in ids["train"]}},
           "test": {SYN: {i: ("test" if k % 2 == 0 else "val") for k, i in enumerate(ids["test"][:-1])}}}  # last id: not in PC15k
    (r2n2 / "ShapeNetRendering").mkdir(parents=True)
    json.dump(split, open(r2n2 / "R2N2_split.json", "w"))
    json.dump(sub, open(r2n2 / "pc_dict_v2.json", "w"))
    for s, v in ids.items():
        for i in v:
            if i in sub[s][SYN]:
                d = root / SYN / sub[s][SYN][i]
                d.mkdir(parents=True, exist_ok=True)
                np.save(d / f"{i}.npy", (rng.standard_normal((n_points, 3)) * 0.2 + [0.1, -0.05, 0.02]).astype(np.float32))
            rd = r2n2 / "ShapeNetRendering" / SYN / i / "rendering"
            rd.mkdir(parents=True)
            Image.fromarray(rng.integers(0, 255, (137, 137, 4), dtype=np.uint8), "RGBA").save(rd / "00.png")
            with open(rd / "rendering_metadata.txt", "w") as f:
                for k in range(24):
                    f.write(f"{15.0 * k + 3.0} {25.0 + (k % 5)} 0 {0.65 + 0.01 * k} 25\n")
    return root, r2n2


def test_r2n2_reader_on_fixture(tmp_path):
    from bdm_amd import datasets as D
    root, r2n2 = make_r2n2_fixture(tmp_path)
    np.random.seed(0)
    ds = D.ShapeNet_R2N2(str(root), str(r2n2), split="test", sample_size=512, img_size=224)
    assert len(ds) == 2                                          # the third id is not in pc_dict: skipped (:413-414)
    s = ds[0]
    assert list(s.keys())[:3] == ["frame_number", "sequence_name", "sequence_category"] and len(s) == 24
    assert s["frame_number"] == "00" and s["sequence_category"] == "chair" and s["sequence_name"].endswith("_00")
    assert s["image_rgb"].shape == (3, 224, 224) and 0.0 <= float(s["image_rgb"].min()) and float(s["image_rgb"].max()) <= 1.0
    assert s["sequence_point_cloud"].shape == (512, 3) and s["fg_probability"] is None
    # dataset-wide normalisation (:467-478): per-axis mean removed; ONE scalar std, taken over all raw values BEFORE the mean
    # is removed (so the normalised values' std is close to, not exactly, 1)
    assert ds.all_points.reshape(-1, 3).mean(0).abs().max() < 1e-4 and abs(float(ds.all_points.reshape(-1).std()) - 1.0) < 0.05
    raw = torch.cat([D.transform_v2_to_v1(torch.tensor(np.load(p)))[None] for p in ds.point_clouds_path])
    assert torch.allclose(ds.all_points_std.reshape(()), raw.reshape(-1).std(), rtol=1e-6)
    # order: random.Random(38383) shuffle of the load order (:446-456)
    import random
    order = list(range(2)); random.Random(38383).shuffle(order)
    assert [p.split("/")[-3] for p in ds.img_path] == [["te0000", "te0001"][i] for i in order]
    batch = D.custom_collate([ds[0], ds[1]])
    assert isinstance(batch["camera"], list) and len(batch["camera"]) == 2 and batch["fg_probability"] is None
    assert batch["image_rgb"].shape == (2, 3, 224, 224) and batch["sequence_point_cloud"].shape == (2, 512, 3)
    assert batch["meta"]["dataset_index"].tolist() == [0, 1]


def test_r2n2_camera_folds_the_normalisation():
    """build_camera_from_R2N2 (shapenet_r2n2.py:65-95) with std = 1: viewing the mean-shifted cloud through the adjusted
    camera equals viewing the raw cloud through the raw R2N2 camera (x, y flipped to PyTorch3D's +X left / +Y up)."""
    from bdm_amd import datasets as D
    from bdm_amd.cameras import r2n2_camera
    Rs, Ts = D.compute_camera_calibration(D.compute_extrinsic_matrix(70.0, 27.0, 0.8 * 1.75))
    mean, std = torch.tensor([0.1, -0.2, 0.05]), torch.tensor(1.0)
    cam = D.build_camera_from_R2N2(Rs.clone(), Ts.clone(), mean, std)
    g = torch.Generator().manual_seed(0)
    p_raw = torch.randn(50, 3, generator=g) * 0.3
    view_raw = p_raw @ Rs + Ts
    view_new = ((p_raw - mean) / std) @ cam.R[0] + cam.T[0]
    assert torch.allclose(view_new, view_raw * torch.tensor([1.0, 1.0, 1.0]), atol=1e-5)
    # the synthetic benchmark camera (cameras.r2n2_camera) is the same construction with mean = 0, std = 1
    ref = r2n2_camera(70.0, 27.0, 0.8 * 1.75)
    cam0 = D.build_camera_from_R2N2(Rs.clone(), Ts.clone(), torch.zeros(3), torch.tensor(1.0))
    assert torch.allclose(cam0.R, ref.R, atol=1e-6) and torch.allclose(cam0.T, ref.T, atol=1e-6)
    assert torch.allclose(cam0.focal_length, torch.tensor([[2.1875, 2.1875]]))


def make_pix3d_fixture(base, n=5):
    from PIL import Image
    root, proc = base / "pix3d", base / "pix3d_processed"
    rng = np.random.default_rng(1)
    entries = []
    for i in range(n):
        model, img = f"model/chair/IKEA_{i}/model.obj", f"img/chair/{i:04d}.png"
        for r in (root, proc):
            (r / model).parent.mkdir(parents=True, exist_ok=True)
            (r / img).parent.mkdir(parents=True, exist_ok=True)
        pts = rng.standard_normal((300, 3)) * 0.2
        with open(proc / model, "w") as f:
            f.write("".join(f"v {a:.6f} {b:.6f} {c:.6f}\n" for a, b, c in pts))
        Image.fromarray(rng.integers(0, 255, (224, 224, 3), dtype=np.uint8), "RGB").save(proc / img)
        entries.append({"img": img, "model": model, "category": "chair", "rot_mat": np.eye(3).tolist(),
                        "trans_mat": [0.0, 0.0, 2.0], "focal_length": 35.0, "img_size": [640, 480], "bbox": [200, 100, 440, 420]})
    entries.append({"img": "img/bed/0000.png", "model": "model/bed/X/model.obj", "category": "bed", "rot_mat": np.eye(3).tolist(),
                    "trans_mat": [0, 0, 2.0], "focal_length": 35.0, "img_size": [640, 480], "bbox": [0, 0, 10, 10]})
    json.dump(entries, open(root / "pix3d.json", "w"))
    return root


def test_pix3d_reader_on_fixture(tmp_path):
    from bdm_amd import datasets as D
    root = make_pix3d_fixture(tmp_path)
    tr, te = D.Pix3D(str(root), split="train"), D.Pix3D(str(root), split="test")
    assert (len(tr), len(te)) == (4, 1)                          # 4 : 1 split of the category's entries in file order
    s = te[0]
    assert s["sequence_name"] == "IKEA_4_0004" and s["sequence_category"] == "chair" and s["frame_number"] == "0004"
    pc = s["sequence_point_cloud"]
    assert pc.shape == (300, 3) and abs(float(pc.reshape(-1).std(unbiased=False)) - 1.0) < 1e-3 and pc.mean(0).abs().max() < 1e-4
    assert s["image_rgb"].shape == (3, 224, 224) and s["image_size_hw"].tolist() == [480, 640]
    cam = s["camera"]
    f, sc = 35.0 * 640 / 32, 224 / 320.0
    assert torch.allclose(cam.focal_length, torch.tensor([[sc * f * 2 / 224] * 2]), rtol=1e-5)   # screen -> NDC (cameras.py)
    # raw branch: area-weighted mesh sampling
    obj = tmp_path / "tri.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 5 5 5\nf 1 2 3\n")
    pts = D.sample_points_from_obj(str(obj), 2000, generator=torch.Generator().manual_seed(0))
    assert pts.shape == (2000, 3) and (pts[:, 2] == 0).all() and (pts[:, 0] + pts[:, 1] <= 1 + 1e-9).all()
    assert abs(pts[:, 0].mean() - 1 / 3) < 0.03


def test_config_accepts_the_recipe_keys():
    from bdm_amd.config import parse_overrides
    cfg = parse_overrides(["dataset=shapenet_r2n2", "dataset.root=/x", "dataset.r2n2_dir=/y", "dataset.which_view_from24=00",
                           "dataset.max_points=4096", "dataloader.batch_size=16", "dataset.category=chair"])
    assert cfg.dataset.type == "shapenet_r2n2" and cfg.dataset.which_view_from24 == "00" and cfg.dataset.root == "/x"


@pytest.mark.gpu
def test_cli_on_r2n2_fixture(hip, tmp_path):
    """main_blending.py with dataset=shapenet_r2n2 on the on-disk fixture: the recipe's dataset path end to end."""
    import main_blending
    from bdm_amd.io import load_pointcloud_ply
    root, r2n2 = make_r2n2_fixture(tmp_path / "data", n_test=4)
    out = main_blending.main(["run.job=sample_bdm_blending", f"run.save_dir={tmp_path / 'out'}", "run.name=r2n2", "dataset=shapenet_r2n2",
                              f"dataset.root={root}", f"dataset.r2n2_dir={r2n2}", "dataset.max_points=1024", "dataloader.batch_size=2",
                              "aux_run.roll_step=1", "aux_run.milestones=[1000,998,996,995]", "run.rng=per_shape"])
    files = sorted(os.listdir(out / "pred" / "chair"))
    assert files == ["te0000_00.ply", "te0001_00.ply", "te0002_00.ply"]     # te0003 is not in pc_dict
    gt = load_pointcloud_ply(out / "gt" / "chair" / files[0])
    assert gt.shape == (1024, 3) and np.isfinite(load_pointcloud_ply(out / "pred" / "chair" / files[0])).all()


def test_batch_sharding_mirrors_accelerate():
    """get_dataset(world > 1) deals whole batches round-robin and pads the tail from the start of the dataset, as
    accelerator.prepare(dataloader) does in the reference (main_blending.py:115-124).  Checked against the installed accelerate."""
    from bdm_amd.datasets import accelerate_batch_shard
    assert accelerate_batch_shard(5, 2, 0, 1) == [[0, 1], [2, 3], [4]]
    # 5 samples, batch 2, 2 ranks: batches (0,1) (2,3) (4) -> rank 0: (0,1), (4,0); rank 1: (2,3), (1,2)
    assert accelerate_batch_shard(5, 2, 0, 2) == [[0, 1], [4, 0]]
    assert accelerate_batch_shard(5, 2, 1, 2) == [[2, 3], [1, 2]]
    acc = pytest.importorskip("accelerate.data_loader")
    import torch.utils.data as tud
    for n, bs, world in [(5, 2, 2), (16, 4, 2), (17, 4, 8), (3, 4, 2), (100, 16, 8), (33, 8, 4), (1, 1, 3), (24, 8, 3)]:
        for rank in range(world):
            base = tud.BatchSampler(tud.SequentialSampler(range(n)), batch_size=bs, drop_last=False)
            want = [list(b) for b in acc.BatchSamplerShard(base, num_processes=world, process_index=rank)]
            assert accelerate_batch_shard(n, bs, rank, world) == want, (n, bs, world, rank)
