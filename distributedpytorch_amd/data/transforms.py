"""Composable joint image/mask transforms on a sample dict — public surface of
`/root/reference/custom_transforms.py` (SURVEY.md §2.2, T1–T14), plus the semantic-segmentation
transforms the north star names (random scale, crop, flip, normalize, to-tensor) and ``Compose``
(torchvision, which the reference uses for composition, is not installed).

Every transform is ``__call__(sample: dict) -> dict``, mutates and returns the dict, and defines
``__str__`` (those strings go into the run report, `/root/reference/train_pascal.py:157-159`).

Behaviour pinned on purpose (tests/test_transforms.py):

* key skipping is by SUBSTRING: a key is left untouched when it contains ``'id'`` or ``'meta'``
  (`custom_transforms.py:108`, `:209`, `:482`).  ``'void_pixels'`` contains ``'id'``, so it is
  skipped by the flip, the scale/rotate and ``ToTensor`` exactly as in the reference.
* ``ScaleNRotate`` in tuple mode draws ``rot`` centred on 0 and ``scale`` centred on 1 from the
  WIDTH of the given ranges (`:93-99`), casts to uint8 before warping (`:124-126`) and picks
  nearest interpolation for arrays whose values are all in {0, 1, 255} (`:117-118`).
* ``FixedResize`` deletes keys that are not listed in ``resolutions`` (`:194-195`).
* ``ToTensor`` only transposes HWC→CHW and casts to float32: no /255, no mean/std (`:489-498`).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np
import numpy.random as random
import torch

from . import helpers, imgproc
from .nellipse import compute_nellipse, compute_nellipse_gaussianHM_fast
from .skewed_axes_weight_map import (generate_mvgauss_image, generate_mvL1L2_image_skewed_axes,
                                     normalize_wtMap)

is_val = False  # module-level switch set by trainValMode (kept for API compatibility; unused here too)


def _skipped(key: str) -> bool:
    return "id" in key or "meta" in key


class Compose:
    """Call each transform in order; exposes ``.transforms`` like torchvision's."""

    def __init__(self, transforms: Sequence):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample

    def __str__(self):
        return "Compose(" + ", ".join(str(t) for t in self.transforms) + ")"


def trainValMode(mode: str) -> None:
    """Sets the module-level ``is_val`` flag (`custom_transforms.py:56-64`); raises instead of
    calling ``exit()`` on bad input."""
    global is_val
    if mode == "train":
        is_val = False
    elif mode == "val":
        is_val = True
    else:
        raise ValueError("Please enter train or val mode")


# --------------------------------------------------------------------------------------
# T1, T2: flip, scale + rotate
# --------------------------------------------------------------------------------------
class RandomHorizontalFlip:
    """One coin per sample (p = 0.5), shared by every non-skipped key."""

    def __call__(self, sample):
        if random.random() < 0.5:
            for key in sample.keys():
                if not _skipped(key):
                    sample[key] = imgproc.flip_horizontal(sample[key])
        return sample

    def __str__(self):
        return "RandomHorizontalFlip"


class ScaleNRotate:
    """``rots``/``scales`` are both tuples ``(lo, hi)`` (continuous) or both lists (discrete)."""

    def __init__(self, rots=(-30, 30), scales=(.75, 1.25), semseg: bool = False):
        if not isinstance(rots, type(scales)):
            raise TypeError("rots and scales must be of the same type")
        self.rots, self.scales, self.semseg = rots, scales, semseg

    def _draw(self) -> Tuple[float, float]:
        if isinstance(self.rots, tuple):
            rw = self.rots[1] - self.rots[0]
            sw = self.scales[1] - self.scales[0]
            return rw * random.random() - rw / 2, sw * random.random() - sw / 2 + 1
        return (self.rots[random.randint(0, len(self.rots))],
                self.scales[random.randint(0, len(self.scales))])

    def __call__(self, sample):
        rot, sc = self._draw()
        for key in list(sample.keys()):
            if _skipped(key):
                continue
            arr = sample[key]
            h, w = arr.shape[:2]
            m = imgproc.get_rotation_matrix_2d((w / 2, h / 2), rot, sc)
            label_like = bool(((arr == 0) | (arr == 1) | (arr == 255)).all())
            if label_like or ("gt" in key and self.semseg):
                interp = imgproc.INTER_NEAREST
            else:
                interp = imgproc.INTER_CUBIC
            border = 255.0 if ("bb_mask" in key or ("gt" in key and self.semseg)) else 0.0
            as_u8 = np.asarray(arr).astype(np.uint8).astype(np.float32)
            out = imgproc.warp_affine(as_u8, m, (w, h), interp, border)
            sample[key] = np.clip(np.rint(out), 0, 255).astype(np.uint8)
        return sample

    def __str__(self):
        return "ScaleNRotate:(rot=" + str(self.rots) + ",scale=" + str(self.scales) + ")"


# --------------------------------------------------------------------------------------
# T3: FixedResize
# --------------------------------------------------------------------------------------
class FixedResize:
    def __init__(self, resolutions: Optional[Dict] = None, flagvals: Optional[Dict] = None):
        self.resolutions, self.flagvals = resolutions, flagvals
        if flagvals is not None and len(resolutions) != len(flagvals):
            raise ValueError("flagvals must match resolutions")

    def _resize(self, arr, key):
        flag = None if self.flagvals is None else self.flagvals[key]
        return helpers.fixed_resize(arr, self.resolutions[key], flagval=flag)

    def __call__(self, sample):
        if self.resolutions is None:
            return sample
        for key in list(sample.keys()):
            if "crop_relax" in key or "meta" in key or "bbox" in key:
                continue
            if "extreme_points_coord" in key:
                if key in self.resolutions:
                    bbox = sample["bbox"]
                    crop_size = np.array([bbox[3] - bbox[1] + 1, bbox[4] - bbox[2] + 1])
                    res = np.array(self.resolutions[key], dtype=np.float32)
                    sample[key] = np.round(sample[key] * res / crop_size).astype(np.int64)
                continue
            if key not in self.resolutions:
                del sample[key]
                continue
            if self.resolutions[key] is None:
                continue
            value = sample[key]
            if isinstance(value, list):
                sample[key] = np.stack([self._resize(crop, key) for crop in value], axis=-1).astype(np.float32)
            else:
                sample[key] = self._resize(value, key)
        return sample

    def __str__(self):
        return "FixedResize:" + str(self.resolutions)


# --------------------------------------------------------------------------------------
# T4, T5: crops from a mask
# --------------------------------------------------------------------------------------
def _crop_all(sample, crop_elems, mask_elem, relax, zero_pad):
    target = sample[mask_elem]
    if target.ndim == 2:
        target = target[..., None]
    for elem in crop_elems:
        img = sample[elem]
        crops = []
        for k in range(target.shape[-1]):
            tk = target[..., k]
            src = (img[..., k] if img.ndim == 3 else img) if elem == mask_elem else img
            if tk.max() == 0:
                crops.append(np.zeros(src.shape, dtype=img.dtype))
            else:
                crops.append(helpers.crop_from_mask(src, tk, relax=relax, zero_pad=zero_pad))
        sample["crop_" + elem] = crops[0] if len(crops) == 1 else crops
    return sample


class CropFromMaskStatic:
    """Crop ``crop_elems`` to the bbox of ``mask_elem`` enlarged by a fixed ``relax``."""

    def __init__(self, crop_elems=("image", "gt"), mask_elem="gt", relax=0, zero_pad=False):
        self.crop_elems, self.mask_elem, self.relax, self.zero_pad = crop_elems, mask_elem, relax, zero_pad

    def __call__(self, sample):
        return _crop_all(sample, self.crop_elems, self.mask_elem, self.relax, self.zero_pad)

    def __str__(self):
        return ("CropFromMask:(crop_elems=" + str(self.crop_elems) + ", mask_elem=" + str(self.mask_elem) +
                ", relax=" + str(self.relax) + ",zero_pad=" + str(self.zero_pad) + ")")


class CropFromMask:
    """Dynamic relax so the object occupies ``dz`` of the ``d``×``d`` network input
    (`custom_transforms.py:377-452`): ``dz = ⌊√(d²/2)⌋`` (val) or uniform in
    ``[⌊√(.45 d²)⌋, ⌊√(.6 d²)⌋)`` per call (train)."""

    def __init__(self, crop_elems=("image", "gt"), mask_elem="gt", zero_pad=False, d=512, is_val=None):
        self.crop_elems, self.mask_elem, self.zero_pad, self.d, self.is_val = crop_elems, mask_elem, zero_pad, d, is_val
        dz = int(math.sqrt(d ** 2 * 0.5))
        self.thresh = ((d - dz) * (d / 5)) / (2 * dz)
        self.dz = float(dz)
        self.min_, self.max_ = int(math.sqrt(d ** 2 * 0.45)), int(math.sqrt(d ** 2 * 0.6))
        self.crop_relax = 0

    def __call__(self, sample):
        if self.is_val is False:
            self.dz = float(np.random.randint(self.min_, self.max_))
        target = sample[self.mask_elem]
        if len(np.unique(target)) == 1:
            sample["crop_image"], sample["crop_gt"] = sample["image"], sample["gt"]
            return sample
        bbox = helpers.get_bbox(target if target.ndim == 2 else target.max(axis=-1))
        side = max(bbox[2] - bbox[0], bbox[3] - bbox[1], 1)
        zoom = self.dz / side
        relax = max((self.d - side * zoom) / (2 * zoom), self.thresh)
        self.crop_relax = int(math.ceil(relax))
        sample["crop_relax"] = self.crop_relax
        return _crop_all(sample, self.crop_elems, self.mask_elem, self.crop_relax, self.zero_pad)

    def __str__(self):
        return ("CropFromMask:(crop_elems=" + str(self.crop_elems) + ", mask_elem=" + str(self.mask_elem) +
                ", relax=dynamic,zero_pad=" + str(self.zero_pad) + ")")


# --------------------------------------------------------------------------------------
# T6–T9
# --------------------------------------------------------------------------------------
class ToImage:
    """Min-max normalise element(s) to ``[0, custom_max]`` (ε = 1e-10)."""

    def __init__(self, norm_elem="image", custom_max=255.):
        self.norm_elem, self.custom_max = norm_elem, custom_max

    def __call__(self, sample):
        elems = self.norm_elem if isinstance(self.norm_elem, tuple) else (self.norm_elem,)
        for e in elems:
            t = sample[e]
            sample[e] = self.custom_max * (t - t.min()) / (t.max() - t.min() + 1e-10)
        return sample

    def __str__(self):
        return "NormalizeImage"


class ToTensor:
    """ndarrays → float32 CHW tensors; skips id/meta/crop_relax keys, passes ``bbox`` through."""

    def __call__(self, sample):
        for key in sample.keys():
            if _skipped(key) or "crop_relax" in key:
                continue
            arr = sample[key]
            if "bbox" in key:
                sample[key] = torch.from_numpy(np.asarray(arr))
                continue
            arr = np.asarray(arr)
            if arr.ndim == 2:
                arr = arr[:, :, None]
            sample[key] = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).float()
        return sample

    def __str__(self):
        return "ToTensor"


class ConcatInputs:
    def __init__(self, elems=("image", "point")):
        self.elems = elems

    def __call__(self, sample):
        first = sample[self.elems[0]]
        parts = [first if first.ndim == 3 else first[..., None]]
        for e in self.elems[1:]:
            t = sample[e]
            if t.shape[:2] != first.shape[:2]:
                raise ValueError(f"ConcatInputs: {e} is {t.shape[:2]}, expected {first.shape[:2]}")
            parts.append(t if t.ndim == 3 else t[..., None])
        sample["concat"] = np.concatenate(parts, axis=2)
        return sample

    def __str__(self):
        return "ExtremePoints:" + str(self.elems)   # (sic) the reference reports this name


class CreateBBMask:
    """``bb_mask`` = 255 outside the gt bounding box, 0 inside."""

    def __call__(self, sample):
        msk = sample["gt"]
        bbox = helpers.get_bbox(msk)
        out = np.full(msk.shape, 255.0, dtype=np.float32)
        if bbox is not None:
            out[bbox[1]:bbox[3], bbox[0]:bbox[2]] = 0
        sample["bb_mask"] = out
        return sample

    def __str__(self):
        return "CreateBBMask"


# --------------------------------------------------------------------------------------
# T10–T13: point / guidance maps
# --------------------------------------------------------------------------------------
def _points(target, pert, val):
    fn = helpers.extreme_points_fixed if val else helpers.extreme_points
    return fn(target, pert)


class ExtremePoints:
    def __init__(self, sigma=10, pert=0, elem="gt", is_val=None):
        self.sigma, self.pert, self.elem, self.is_val = sigma, pert, elem, is_val

    def __call__(self, sample):
        target = sample[self.elem]
        if target.ndim == 3:
            raise ValueError("ExtremePoints not implemented for multiple object per image.")
        if target.max() == 0:
            sample["extreme_points"] = np.zeros(target.shape, dtype=target.dtype)
        else:
            pts = _points(target, self.pert, bool(self.is_val))
            sample["extreme_points"] = helpers.make_gt(target, pts, sigma=self.sigma, one_mask_per_point=False)
        return sample

    def __str__(self):
        return "ExtremePoints:(sigma=" + str(self.sigma) + ", pert=" + str(self.pert) + ", elem=" + str(self.elem) + ")"


class NEllipse:
    def __init__(self, is_val=True):
        self.is_val = is_val

    def __call__(self, sample):
        target = sample["crop_gt"]
        if target.max() == 0:
            sample["nellipse"] = np.zeros(target.shape, dtype=target.dtype)
        else:
            pts = _points(target, 0, self.is_val is True)
            sample["nellipse"] = compute_nellipse(np.arange(target.shape[0]), np.arange(target.shape[1]), pts) * 255
        return sample

    def __str__(self):
        return "NEllipse"


class NEllipseWithGaussians:
    """``z = 255·z1 + alpha·255·z2`` rescaled to a maximum of 255 — the guidance channel the
    reference script trains with (`train_pascal.py:131`, alpha 0.6)."""

    def __init__(self, alpha=0.6, is_val=True):
        self.alpha, self.is_val = alpha, is_val

    def __call__(self, sample):
        target = sample["crop_gt"]
        if target.max() == 0:
            sample["nellipseWithGaussians"] = np.zeros(target.shape, dtype=target.dtype)
        else:
            pts = _points(target, 0, self.is_val is True)
            z1, z2 = compute_nellipse_gaussianHM_fast(np.arange(target.shape[0]), np.arange(target.shape[1]), pts)
            z = 255.0 * z1 + self.alpha * 255.0 * z2
            # min(): z·(255/max) can land one float32 ulp above 255 and trip the 0..255 batch assert (train_pascal.py:189)
            z = (z * (255.0 / max(float(z.max()), 1e-10))).astype(np.float32)
            sample["nellipseWithGaussians"] = np.minimum(z, np.float32(255.0), out=z)
        return sample

    def __str__(self):
        return "NEllipseWithGaussians:(alpha=" + str(self.alpha) + ")"


class AddConfidenceMap:
    """Append a confidence channel (``l1l2`` skewed-axes or ``gaussian``) → ``with_hm``."""

    def __init__(self, elem="image", hm_type="l1l2", tau=1, pert=0, is_val=None):
        self.elem, self.hm_type, self.tau, self.pert, self.is_val = elem, hm_type, tau, pert, is_val

    def __call__(self, sample):
        img = sample[self.elem]
        msk = sample["crop_gt"].astype(bool)
        if len(np.unique(msk)) == 1:
            hm = np.zeros(img.shape[:2])
        else:
            if self.hm_type == "l1l2":
                pts = _points(msk.astype(np.float32), self.pert, bool(self.is_val))
                h_map, _, _ = generate_mvL1L2_image_skewed_axes(msk, extreme_points=pts, FULL_IMAGE_WEIGHTS=1,
                                                                d2_THRESH=None, tau=self.tau)
            elif self.hm_type == "gaussian":
                h_map = generate_mvgauss_image(msk, FULL_IMAGE_WEIGHTS=1, tau=0.5)
            else:
                raise ValueError(f"unknown hm_type {self.hm_type}")
            hm = normalize_wtMap(h_map) * 255
        sample["with_hm"] = np.concatenate([img, hm[..., None]], axis=2).astype(np.float64)
        return sample

    def __str__(self):
        return "AddConfidenceMap:(type=" + str(self.hm_type) + ", tau=" + str(self.tau) + ")"


# --------------------------------------------------------------------------------------
# semantic-segmentation transforms (north star: random scale, crop, flip, normalize, to-tensor)
# --------------------------------------------------------------------------------------
class RandomScaleCrop:
    """DeepLab training augmentation on ``image`` (bilinear) + ``gt`` (nearest, 255 padding):
    random isotropic scale in ``scales``, pad to ``crop_size`` if smaller, random crop."""

    def __init__(self, crop_size=513, scales=(0.5, 2.0), fill_label=255, fill_image=(124.0, 116.0, 104.0)):
        self.crop = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
        self.scales, self.fill_label, self.fill_image = scales, fill_label, fill_image

    def __call__(self, sample):
        img, gt = np.asarray(sample["image"], dtype=np.float32), np.asarray(sample["gt"], dtype=np.float32)
        h, w = gt.shape[:2]
        s = random.uniform(self.scales[0], self.scales[1])
        nh, nw = max(int(round(h * s)), 1), max(int(round(w * s)), 1)
        img = imgproc.resize(img, (nh, nw), imgproc.INTER_LINEAR)
        gt = imgproc.resize(gt, (nh, nw), imgproc.INTER_NEAREST)
        ch, cw = self.crop
        ph, pw = max(ch - nh, 0), max(cw - nw, 0)
        if ph or pw:
            canvas = np.empty((nh + ph, nw + pw, 3), dtype=np.float32)
            canvas[...] = np.asarray(self.fill_image, dtype=np.float32)
            canvas[:nh, :nw] = img
            lab = np.full((nh + ph, nw + pw), float(self.fill_label), dtype=np.float32)
            lab[:nh, :nw] = gt
            img, gt = canvas, lab
        y0 = random.randint(0, img.shape[0] - ch + 1)
        x0 = random.randint(0, img.shape[1] - cw + 1)
        sample["image"], sample["gt"] = img[y0:y0 + ch, x0:x0 + cw], gt[y0:y0 + ch, x0:x0 + cw]
        return sample

    def __str__(self):
        return "RandomScaleCrop:(crop=" + str(self.crop) + ",scales=" + str(self.scales) + ")"


class FixedScaleCenterCrop:
    """Validation counterpart: resize the short side to ``crop_size`` and centre-crop / pad."""

    def __init__(self, crop_size=513, fill_label=255):
        self.crop = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
        self.fill_label = fill_label

    def __call__(self, sample):
        img, gt = np.asarray(sample["image"], dtype=np.float32), np.asarray(sample["gt"], dtype=np.float32)
        ch, cw = self.crop
        h, w = gt.shape[:2]
        s = max(ch / h, cw / w)
        nh, nw = max(int(round(h * s)), ch), max(int(round(w * s)), cw)
        img = imgproc.resize(img, (nh, nw), imgproc.INTER_LINEAR)
        gt = imgproc.resize(gt, (nh, nw), imgproc.INTER_NEAREST)
        y0, x0 = (nh - ch) // 2, (nw - cw) // 2
        sample["image"], sample["gt"] = img[y0:y0 + ch, x0:x0 + cw], gt[y0:y0 + ch, x0:x0 + cw]
        return sample

    def __str__(self):
        return "FixedScaleCenterCrop:(crop=" + str(self.crop) + ")"


class Normalize:
    """``(x − mean) / std`` per channel on ``elem`` (values in 0..255 by default)."""

    def __init__(self, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), elem="image"):
        self.mean, self.std, self.elem = np.asarray(mean, np.float32), np.asarray(std, np.float32), elem

    def __call__(self, sample):
        sample[self.elem] = (np.asarray(sample[self.elem], dtype=np.float32) - self.mean) / self.std
        return sample

    def __str__(self):
        return "Normalize:(mean=" + str(tuple(self.mean.tolist())) + ",std=" + str(tuple(self.std.tolist())) + ")"


class SemSegToTensor:
    """``image`` → float32 CHW tensor, ``gt`` → int64 HW class map (255 = ignore)."""

    def __call__(self, sample):
        img = np.asarray(sample["image"], dtype=np.float32)
        sample["image"] = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        sample["gt"] = torch.from_numpy(np.ascontiguousarray(np.asarray(sample["gt"]))).long()
        return sample

    def __str__(self):
        return "SemSegToTensor"
