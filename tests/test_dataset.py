"""Dataset layer against a tiny synthetic VOC tree in tmp_path."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from distributedpytorch_amd.data import (CombineDBs, SyntheticVOC, VOCSegmentation, helpers,
                                         write_synthetic_voc_tree)
from distributedpytorch_amd.data import transforms as tr


@pytest.fixture(scope="module")
def voc_root(tmp_path_factory):
    return write_synthetic_voc_tree(str(tmp_path_factory.mktemp("voc")), n_train=5, n_val=3, size=(72, 96), seed=3)


def test_instance_dataset_contract(voc_root):
    ds = VOCSegmentation(root=voc_root, split="train", check_integrity=False)
    assert len(ds) == len(ds.obj_list) > 0
    s = ds[0]
    assert set(s.keys()) == {"image", "gt", "void_pixels", "meta"}
    assert s["image"].dtype == np.float32 and s["image"].shape == (72, 96, 3)
    assert 0 <= s["image"].min() and s["image"].max() <= 255
    assert s["gt"].dtype == np.float32 and set(np.unique(s["gt"])) <= {0.0, 1.0}
    assert set(np.unique(s["void_pixels"])) <= {0.0, 1.0}
    assert not np.any((s["gt"] > 0) & (s["void_pixels"] > 0))         # void zeroed in gt
    assert set(s["meta"].keys()) == {"image", "object", "category", "im_size"}
    assert s["meta"]["im_size"] == (72, 96) and 1 <= s["meta"]["category"] <= 20
    assert str(ds) == "VOC2012(split=['train'],area_thres=0)"
    assert ds.category_names[0] == "background" and len(ds.category_names) == 21


def test_preprocess_cache_roundtrip_and_area_threshold(voc_root):
    ds = VOCSegmentation(root=voc_root, split=["val", "train"], check_integrity=False)
    assert ds.split == ["train", "val"]                               # sorted like the reference
    assert os.path.basename(ds.obj_list_file) == "train_val_instances.txt"
    with open(ds.obj_list_file) as fh:
        cached = json.load(fh)
    assert sorted(cached.keys()) == sorted(ds.im_ids)
    # every listed object's category equals the class map at the object's first pixel
    ii, jj = ds.obj_list[0]
    inst = np.array(Image.open(ds.masks[ii]))
    cls = np.array(Image.open(ds.categories[ii]))
    ys, xs = np.nonzero(inst == jj + 1)
    assert cached[ds.im_ids[ii]][jj] == int(cls[ys[0], xs[0]])
    # second construction re-uses the cache (mtime unchanged)
    mt = os.path.getmtime(ds.obj_list_file)
    VOCSegmentation(root=voc_root, split=["train", "val"], check_integrity=False)
    assert os.path.getmtime(ds.obj_list_file) == mt
    big = VOCSegmentation(root=voc_root, split="train", area_thres=10 ** 6, check_integrity=False)
    assert len(big) == 0 and big.obj_list_file.endswith("train_instances_area_thres-1000000.txt")


def test_default_mode_returns_instance_ids(voc_root):
    ds = VOCSegmentation(root=voc_root, split="val", default=True, retname=False, check_integrity=False)
    s = ds[0]
    assert "meta" not in s and s["gt"].max() >= 1 and 255 not in np.unique(s["gt"])


def test_semantic_mode(voc_root):
    ds = VOCSegmentation(root=voc_root, split="val", semantic=True, check_integrity=False)
    assert len(ds) == 3
    s = ds[1]
    assert set(np.unique(s["gt"]).astype(int)) <= set(range(21)) | {255}
    assert 255 in np.unique(s["gt"])                                   # object borders are void


def test_integrity_check_is_enforced_by_default(voc_root):
    with pytest.raises(RuntimeError):
        VOCSegmentation(root=voc_root, split="val")                    # no tar, MD5 cannot match


def test_missing_file_is_reported(voc_root, tmp_path):
    import shutil
    broken = str(tmp_path / "broken")
    shutil.copytree(voc_root, broken)
    voc = os.path.join(broken, VOCSegmentation.BASE_DIR)
    victim = sorted(os.listdir(os.path.join(voc, "SegmentationObject")))[0]
    os.remove(os.path.join(voc, "SegmentationObject", victim))
    with pytest.raises(FileNotFoundError):
        VOCSegmentation(root=broken, split=["train", "val"], check_integrity=False)


def test_dataloader_batches_through_reference_pipeline(voc_root):
    np.random.seed(0)
    pipe = tr.Compose([tr.RandomHorizontalFlip(), tr.ScaleNRotate(rots=(-20, 20), scales=(.75, 1.25)),
                       tr.CropFromMaskStatic(crop_elems=("image", "gt"), relax=8, zero_pad=True),
                       tr.FixedResize(resolutions={"crop_image": (32, 32), "crop_gt": (32, 32)}),
                       tr.NEllipseWithGaussians(alpha=0.6, is_val=False),
                       tr.ConcatInputs(elems=("crop_image", "nellipseWithGaussians")), tr.ToTensor()])
    ds = VOCSegmentation(root=voc_root, split="train", transform=pipe, check_integrity=False)
    dl = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=True, drop_last=True)
    b = next(iter(dl))
    assert b["concat"].shape == (2, 4, 32, 32) and b["crop_gt"].shape == (2, 1, 32, 32)
    assert b["concat"].min() >= 0 and b["concat"].max() <= 255


def test_synthetic_voc_and_combine(voc_root):
    a = SyntheticVOC(6, (33, 33), mode="semantic", seed=1)
    s0, s0b = a[0], a[0]
    assert np.array_equal(s0["image"], s0b["image"]) and s0["image"].shape == (33, 33, 3)   # deterministic
    assert set(np.unique(s0["gt"]).astype(int)) <= set(range(21)) | {255}
    inst = SyntheticVOC(4, (33, 33), mode="instance")[1]
    assert set(inst.keys()) == {"image", "gt", "void_pixels", "meta"}
    tr_db = VOCSegmentation(root=voc_root, split="train", check_integrity=False)
    both = VOCSegmentation(root=voc_root, split=["train", "val"], check_integrity=False)
    val_db = VOCSegmentation(root=voc_root, split="val", check_integrity=False)
    comb = CombineDBs([tr_db, both], excluded=[val_db])
    assert len(comb) == len(tr_db)                                    # val excluded, train not duplicated
    assert comb[0]["meta"]["image"] in tr_db.im_ids


def test_helpers_geometry_and_metrics():
    m = np.zeros((20, 30), np.float32)
    m[5:9, 10:16] = 1
    assert helpers.get_bbox(m) == (10, 5, 15, 8)
    assert helpers.get_bbox(m, pad=50, zero_pad=False) == (0, 0, 29, 19)
    assert helpers.get_bbox(m, pad=7, zero_pad=True) == (3, -2, 22, 15)
    assert helpers.get_bbox(np.zeros((4, 4))) is None
    img = np.arange(20 * 30, dtype=np.float32).reshape(20, 30)
    c = helpers.crop_from_mask(img, m, relax=7, zero_pad=True)
    assert c.shape == (18, 20) and c[0, 0] == 0 and c[2, 0] == img[0, 3]
    assert helpers.calc_jaccard(m, m) == 1.0 and helpers.calc_jaccard(np.zeros_like(m), np.zeros_like(m)) == 1.0
    half = m.copy()
    half[:, 13:] = 0
    assert abs(helpers.calc_jaccard(half, m) - 0.5) < 1e-6
    void = np.zeros_like(m)
    void[:, 13:] = 1
    assert helpers.calc_jaccard(half, m, void) == 1.0
    # crop → resize → paste back reproduces the mask
    bbox = helpers.get_bbox(m, pad=3, zero_pad=True)
    crop = helpers.fixed_resize(helpers.crop_from_mask(m, m, relax=3, zero_pad=True), (64, 64))
    full = helpers.crop2fullmask(crop, bbox, m, zero_pad=True, relax=0)
    assert helpers.calc_jaccard(full > 0.5, m) > 0.9
    t = torch.arange(24.).view(1, 2, 3, 4)
    assert helpers.tens2image(t).shape == (3, 4, 2)
    rgba = helpers.colorMaskWithAlpha(m, color="r", transparency=0.3)
    assert rgba.shape == (20, 30, 4) and rgba[6, 12].tolist() == [1.0, 0.0, 0.0, pytest.approx(0.3)]
    ov = helpers.overlay_mask(np.ones((20, 30, 3), np.float32), m)
    assert ov.shape == (20, 30, 3)
    g = helpers.make_gt(m, [[12, 6], [3, 3]], sigma=5)
    assert g[6, 12] == pytest.approx(1.0) and g.shape == (20, 30)


def test_param_report(tmp_path):
    path = str(tmp_path / "sub" / "exp.txt")
    helpers.generate_param_report(path, {"lr": 5e-8, "trainBatch": 16})
    assert open(path).read() == "lr: 5e-08\ntrainBatch: 16\n"
