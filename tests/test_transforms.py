"""Transform layer: determinism under seed, joint image/mask consistency,
dtype/shape contract, FixedResize key deletion, the ``'id' in 'void_pixels'`` quirk, ``__str__``."""
import numpy as np
import pytest
import torch

from distributedpytorch_amd.data import helpers, imgproc
from distributedpytorch_amd.data import transforms as tr


def _sample(h=48, w=64, seed=0):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.float32)
    gt = np.zeros((h, w), np.float32)
    gt[12:30, 20:45] = 1
    void = np.zeros((h, w), np.float32)
    void[11, 20:45] = 1
    return {"image": img, "gt": gt, "void_pixels": void, "meta": {"image": "x", "object": "0"}}


def test_flip_twice_is_identity_and_joint():
    s = _sample()
    ref = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s.items()}
    np.random.seed(0)
    flips = 0
    t = tr.RandomHorizontalFlip()
    for _ in range(40):
        before = s["image"].copy()
        s = t(s)
        if not np.array_equal(before, s["image"]):
            flips += 1
            # image and gt flipped together
            assert np.array_equal(s["gt"], imgproc.flip_horizontal(ref["gt"])) or np.array_equal(s["gt"], ref["gt"])
    assert 5 < flips < 35
    if flips % 2 == 0:
        assert np.array_equal(s["image"], ref["image"]) and np.array_equal(s["gt"], ref["gt"])
    else:
        assert np.array_equal(imgproc.flip_horizontal(s["image"]), ref["image"])


def test_void_pixels_substring_quirk_is_pinned():
    """'id' in 'void_pixels' is True, so flip / scale-rotate / ToTensor leave it untouched
    (custom_transforms.py:108, :209, :482 of the reference)."""
    np.random.seed(3)
    s = _sample()
    void0 = s["void_pixels"].copy()
    for _ in range(8):
        s = tr.RandomHorizontalFlip()(s)
    s = tr.ScaleNRotate(rots=(-20, 20), scales=(.75, 1.25))(s)
    assert np.array_equal(s["void_pixels"], void0)
    s = tr.ToTensor()(s)
    assert isinstance(s["void_pixels"], np.ndarray)      # not converted
    assert torch.is_tensor(s["image"]) and torch.is_tensor(s["gt"])
    assert isinstance(s["meta"], dict)


def test_scale_n_rotate_deterministic_and_joint():
    a, b = _sample(), _sample()
    np.random.seed(7)
    a = tr.ScaleNRotate(rots=(-20, 20), scales=(.75, 1.25))(a)
    np.random.seed(7)
    b = tr.ScaleNRotate(rots=(-20, 20), scales=(.75, 1.25))(b)
    assert np.array_equal(a["image"], b["image"]) and np.array_equal(a["gt"], b["gt"])
    assert a["image"].dtype == np.uint8 and a["image"].shape == (48, 64, 3)
    assert set(np.unique(a["gt"])) <= {0, 1}           # nearest interpolation for the mask
    # same affine for image and mask: warp a mask-coloured image and compare supports
    s = _sample()
    s["image"] = np.repeat(s["gt"][..., None] * 255, 3, axis=2)
    np.random.seed(11)
    s = tr.ScaleNRotate(rots=[15], scales=[1.1])(s)
    agree = ((s["image"][..., 0] > 127) == (s["gt"] > 0)).mean()
    assert agree > 0.97


def test_scale_n_rotate_range_semantics():
    """Tuple mode is centred on 0 / 1 using the WIDTH of the ranges (reference :93-99)."""
    t = tr.ScaleNRotate(rots=(10, 30), scales=(1.0, 1.5))
    np.random.seed(0)
    draws = np.array([t._draw() for _ in range(500)])
    assert -10 <= draws[:, 0].min() and draws[:, 0].max() <= 10
    assert 0.75 <= draws[:, 1].min() and draws[:, 1].max() <= 1.25
    with pytest.raises(TypeError):
        tr.ScaleNRotate(rots=(1, 2), scales=[1.0])


def test_crop_resize_concat_pipeline_contract():
    np.random.seed(0)
    pipe = tr.Compose([
        tr.CropFromMaskStatic(crop_elems=("image", "gt"), relax=5, zero_pad=True),
        tr.FixedResize(resolutions={"crop_image": (32, 32), "crop_gt": (32, 32)}),
        tr.NEllipseWithGaussians(alpha=0.6, is_val=True),
        tr.ConcatInputs(elems=("crop_image", "nellipseWithGaussians")),
        tr.ToTensor()])
    s = pipe(_sample())
    # FixedResize deleted everything not listed (reference :194-195) except meta
    assert set(s.keys()) == {"meta", "crop_image", "crop_gt", "nellipseWithGaussians", "concat"}
    assert s["concat"].shape == (4, 32, 32) and s["concat"].dtype == torch.float32
    assert s["crop_gt"].shape == (1, 32, 32)
    # the in-loop asserts of the reference (train_pascal.py:188-190)
    assert 0 <= float(s["concat"][:3].min()) and float(s["concat"][:3].max()) <= 255
    assert 0 <= float(s["concat"][3].min()) and abs(float(s["concat"][3].max()) - 255) < 1e-3
    assert set(s["crop_gt"].unique().tolist()) == {0.0, 1.0}


def test_fixed_resize_none_keeps_native_size_and_lists_stack():
    s = {"gt": np.ones((10, 12), np.float32), "crop_gt": [np.ones((5, 6), np.float32), np.zeros((7, 3), np.float32)],
         "junk": np.zeros((3, 3), np.float32), "meta": {}}
    out = tr.FixedResize(resolutions={"gt": None, "crop_gt": (8, 8)})(s)
    assert out["gt"].shape == (10, 12) and out["crop_gt"].shape == (8, 8, 2) and "junk" not in out


def test_crop_from_mask_zero_pad_and_dynamic():
    s = _sample()
    out = tr.CropFromMaskStatic(relax=20, zero_pad=True)(dict(s))
    assert out["crop_gt"].shape == (18 + 40, 25 + 40)          # bbox extent + 2·relax, beyond the image
    out2 = tr.CropFromMaskStatic(relax=20, zero_pad=False)(dict(s))
    assert out2["crop_gt"].shape[0] <= 48 and out2["crop_gt"].shape[1] <= 64
    dyn = tr.CropFromMask(d=64, is_val=True, zero_pad=True)(dict(s))
    assert dyn["crop_relax"] >= 1 and dyn["crop_image"].shape[:2] == dyn["crop_gt"].shape[:2]
    empty = dict(s, gt=np.zeros_like(s["gt"]))
    assert tr.CropFromMaskStatic(relax=5)(empty)["crop_gt"].max() == 0


def test_point_and_confidence_maps():
    s = _sample()
    s["crop_gt"], s["crop_image"] = s["gt"], s["image"]
    pts = helpers.extreme_points_fixed(s["gt"], 0)
    assert pts.shape == (4, 2)
    assert pts[0, 0] == 20 and pts[1, 0] == 44 and pts[2, 1] == 12 and pts[3, 1] == 29
    e = tr.ExtremePoints(sigma=10, pert=0, elem="gt", is_val=True)(dict(s))["extreme_points"]
    assert e.shape == (48, 64) and abs(e.max() - 1) < 1e-5
    n = tr.NEllipse(is_val=True)(dict(s))["nellipse"]
    assert n.max() == 255 and n[20, 32] > n[2, 2]                # high inside the object
    for hm in ("l1l2", "gaussian"):
        w = tr.AddConfidenceMap(elem="crop_image", hm_type=hm, tau=7, is_val=True)(dict(s))["with_hm"]
        assert w.shape == (48, 64, 4) and 0 <= w[..., 3].min() and w[..., 3].max() <= 255 + 1e-3
    bb = tr.CreateBBMask()(dict(s))["bb_mask"]
    assert bb[0, 0] == 255 and bb[15, 25] == 0
    im = tr.ToImage(norm_elem="image", custom_max=1.0)(dict(s))["image"]
    assert im.min() == 0 and abs(im.max() - 1) < 1e-6
    with pytest.raises(ValueError):
        tr.ExtremePoints()(dict(s, gt=np.zeros((4, 4, 2))))


def test_semseg_transforms():
    rng = np.random.default_rng(0)
    s = {"image": rng.integers(0, 256, (60, 80, 3)).astype(np.float32),
         "gt": rng.integers(0, 21, (60, 80)).astype(np.float32)}
    np.random.seed(1)
    out = tr.Compose([tr.RandomHorizontalFlip(), tr.RandomScaleCrop(65, (0.5, 2.0)), tr.Normalize(),
                      tr.SemSegToTensor()])(dict(s))
    assert out["image"].shape == (3, 65, 65) and out["gt"].shape == (65, 65) and out["gt"].dtype == torch.int64
    assert set(out["gt"].unique().tolist()) <= set(range(21)) | {255}
    np.random.seed(5)
    small = tr.RandomScaleCrop(129, (0.5, 0.5))(dict(s))       # always needs padding -> 255 ignore
    assert (small["gt"] == 255).any() and small["gt"].shape == (129, 129)
    val = tr.FixedScaleCenterCrop(33)(dict(s))
    assert val["image"].shape == (33, 33, 3)


def test_strings_match_reference_formats():
    assert str(tr.RandomHorizontalFlip()) == "RandomHorizontalFlip"
    assert str(tr.ScaleNRotate(rots=(-20, 20), scales=(.75, 1.25))) == "ScaleNRotate:(rot=(-20, 20),scale=(0.75, 1.25))"
    assert str(tr.FixedResize(resolutions={"a": (1, 2)})) == "FixedResize:{'a': (1, 2)}"
    assert str(tr.ToTensor()) == "ToTensor" and str(tr.ToImage()) == "NormalizeImage"
    assert str(tr.CropFromMaskStatic(relax=50, zero_pad=True)) == \
        "CropFromMask:(crop_elems=('image', 'gt'), mask_elem=gt, relax=50,zero_pad=True)"
    assert str(tr.ExtremePoints(sigma=10, pert=0, elem="gt")) == "ExtremePoints:(sigma=10, pert=0, elem=gt)"
    assert str(tr.ConcatInputs(elems=("a", "b"))) == "ExtremePoints:('a', 'b')"
    c = tr.Compose([tr.ToTensor()])
    assert len(c.transforms) == 1
    tr.trainValMode("val")
    assert tr.is_val is True
    with pytest.raises(ValueError):
        tr.trainValMode("bogus")


def test_imgproc_native_matches_numpy_and_basics():
    rng = np.random.default_rng(0)
    a = (rng.random((23, 31, 3)) * 255).astype(np.float32)
    m = imgproc.get_rotation_matrix_2d((15.5, 11.5), 17.0, 0.9)
    for interp in (imgproc.INTER_NEAREST, imgproc.INTER_LINEAR, imgproc.INTER_CUBIC):
        c = imgproc.warp_affine(a, m, (31, 23), interp, 7.0)
        n = imgproc.warp_affine(a, m, (31, 23), interp, 7.0, force_numpy=True)
        assert np.abs(c - n).max() < 1e-3
        assert np.abs(imgproc.resize(a, (40, 50), interp) - imgproc.resize(a, (40, 50), interp, force_numpy=True)).max() < 1e-3
    ident = imgproc.warp_affine(a, np.array([[1, 0, 0], [0, 1, 0]], float), (31, 23), imgproc.INTER_CUBIC)
    assert np.abs(ident - a).max() < 1e-3
    const = np.full((9, 9), 5.0, np.float32)
    assert np.allclose(imgproc.resize(const, (20, 13), imgproc.INTER_CUBIC), 5.0, atol=1e-4)
    assert np.array_equal(imgproc.flip_horizontal(imgproc.flip_horizontal(a)), a)
