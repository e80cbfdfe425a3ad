"""``dataloaders.helpers`` — the helper surface the reference imports but does not ship
(SURVEY.md §2.4).  Contracts are recovered from the call sites; each function cites them.

Everything here is CPU / NumPy; the GPU twins live in ``ops`` (augmentation, confusion matrix).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import imgproc


# ----------------------------------------------------------------------------- tensors / images
def tens2image(im) -> np.ndarray:
    """CHW (or 1×CHW) tensor → squeezed HWC ndarray (`/root/reference/pascal.py:283-284`,
    `train_pascal.py:286`, `:288`)."""
    if torch.is_tensor(im):
        im = im.detach().cpu().numpy()
    tmp = np.squeeze(np.asarray(im))
    if tmp.ndim == 2:
        return tmp
    return tmp.transpose((1, 2, 0))


def overlay_mask(im: np.ndarray, ma: np.ndarray, colors=None, alpha: float = 0.5) -> np.ndarray:
    """Alpha-blend a binary mask over an image in 0..1 (`pascal.py:283`)."""
    im = np.asarray(im, dtype=np.float32)
    ma = np.asarray(ma) > 0.5
    color = np.array((0.0, 0.7, 0.0) if colors is None else colors, dtype=np.float32)
    if im.ndim == 2:
        im = np.repeat(im[..., None], 3, axis=2)
    out = im.copy()
    out[ma] = (1 - alpha) * im[ma] + alpha * color
    # thin contour: mask pixels with a background 4-neighbour
    pad = np.pad(ma, 1)
    edge = ma & ~(pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:])
    out[edge] = 0.0
    return out


def colorMaskWithAlpha(mask: np.ndarray, color="r", transparency: float = 0.5) -> np.ndarray:
    """Binary mask → RGBA overlay image for ``plt.imshow`` (`train_pascal.py:265`)."""
    table = {"r": (1, 0, 0), "g": (0, 1, 0), "b": (0, 0, 1), "y": (1, 1, 0), "w": (1, 1, 1)}
    rgb = table.get(color, color) if isinstance(color, str) else color
    m = (np.asarray(mask) > 0.5).astype(np.float32)
    out = np.zeros(m.shape + (4,), dtype=np.float32)
    for k in range(3):
        out[..., k] = rgb[k] * m
    out[..., 3] = m * transparency
    return out


# ----------------------------------------------------------------------------- boxes / crops
def get_bbox(mask: np.ndarray, points=None, pad: int = 0, zero_pad: bool = False) -> Optional[Tuple[int, int, int, int]]:
    """Inclusive ``(x_min, y_min, x_max, y_max)`` of the non-zero region, enlarged by ``pad``;
    clipped to the image unless ``zero_pad`` (`custom_transforms.py:70-72`, `:416-417`;
    `train_pascal.py:287`).  ``None`` for an empty mask."""
    if points is not None:
        inds = np.flip(np.asarray(points).T, axis=0)
    else:
        inds = np.where(np.asarray(mask) > 0)
    if inds[0].size == 0:
        return None
    h, w = mask.shape[:2]
    if zero_pad:
        x_lo, y_lo, x_hi, y_hi = -np.inf, -np.inf, np.inf, np.inf
    else:
        x_lo, y_lo, x_hi, y_hi = 0, 0, w - 1, h - 1
    x_min = int(max(inds[1].min() - pad, x_lo))
    y_min = int(max(inds[0].min() - pad, y_lo))
    x_max = int(min(inds[1].max() + pad, x_hi))
    y_max = int(min(inds[0].max() + pad, y_hi))
    return x_min, y_min, x_max, y_max


def crop_from_bbox(img: np.ndarray, bbox: Sequence[int], zero_pad: bool = False) -> np.ndarray:
    """Crop an inclusive bbox; parts outside the image are zero-filled when ``zero_pad``."""
    h, w = img.shape[:2]
    x0, y0, x1, y1 = bbox
    vx0, vy0, vx1, vy1 = max(x0, 0), max(y0, 0), min(x1, w - 1), min(y1, h - 1)
    if zero_pad:
        out = np.zeros((y1 - y0 + 1, x1 - x0 + 1) + img.shape[2:], dtype=img.dtype)
        if vx1 >= vx0 and vy1 >= vy0:
            out[vy0 - y0:vy1 - y0 + 1, vx0 - x0:vx1 - x0 + 1] = img[vy0:vy1 + 1, vx0:vx1 + 1]
        return out
    return img[vy0:vy1 + 1, vx0:vx1 + 1].copy()


def crop_from_mask(img: np.ndarray, mask: np.ndarray, relax: int = 0, zero_pad: bool = False) -> np.ndarray:
    """Crop ``img`` to the bbox of ``mask`` enlarged by ``relax`` px (`custom_transforms.py:359`,
    `:366`, `:436`, `:443`)."""
    if mask.shape[:2] != img.shape[:2]:
        mask = imgproc.resize(mask.astype(np.float32), img.shape[:2], imgproc.INTER_NEAREST)
    bbox = get_bbox(mask, pad=int(relax), zero_pad=zero_pad)
    if bbox is None:
        return np.zeros(img.shape, dtype=img.dtype)
    return crop_from_bbox(img, bbox, zero_pad)


def fixed_resize(sample: np.ndarray, resolution, flagval=None) -> np.ndarray:
    """Resize to ``(h, w)`` (or scale the short side when an int).  Interpolation is nearest for
    label-like arrays (values ⊂ {0, 1, 255}) and bicubic otherwise unless ``flagval`` is given
    (`custom_transforms.py:186-193`)."""
    arr = np.asarray(sample, dtype=np.float32)
    if flagval is None:
        is_label = bool(((arr == 0) | (arr == 1) | (arr == 255)).all())
        flagval = imgproc.INTER_NEAREST if is_label else imgproc.INTER_CUBIC
    if isinstance(resolution, (int, np.integer)):
        h, w = arr.shape[:2]
        s = float(resolution) / min(h, w)
        resolution = (int(round(h * s)), int(round(w * s)))
    out = imgproc.resize(arr, tuple(int(v) for v in resolution), flagval)
    if flagval == imgproc.INTER_CUBIC and arr.size:
        # bicubic overshoots; keep the value range of the input so the 0..255 data contract the
        # reference asserts on every batch (train_pascal.py:188-189) holds for float images too
        np.clip(out, arr.min(), arr.max(), out=out)
    return out


def crop2fullmask(crop_mask: np.ndarray, bbox, im=None, im_size=None, zero_pad: bool = False, relax: int = 0,
                  mask_relax: bool = True, interpolation=imgproc.INTER_CUBIC, scikit: bool = False) -> np.ndarray:
    """Paste a fixed-size crop prediction back into a full-size canvas (`train_pascal.py:290`):
    the crop is resized to the bbox extent and written at the bbox; with ``mask_relax`` the
    ``relax`` border of the crop is suppressed first."""
    if im is None and im_size is None:
        raise ValueError("give im or im_size")
    im_si = im.shape[:2] if im is not None else tuple(im_size)
    x0, y0, x1, y1 = bbox
    bh, bw = y1 - y0 + 1, x1 - x0 + 1
    if bh <= 0 or bw <= 0:
        return np.zeros(im_si, dtype=np.float32)
    crop = imgproc.resize(np.asarray(crop_mask, dtype=np.float32), (bh, bw), interpolation)
    if mask_relax and relax > 0:
        keep = np.zeros_like(crop)
        keep[relax:bh - relax, relax:bw - relax] = 1.0
        # soft floor so thresholds in (0,1) see "background" outside the tight box
        crop = crop * keep
    out = np.zeros(im_si, dtype=np.float32)
    vx0, vy0, vx1, vy1 = max(x0, 0), max(y0, 0), min(x1, im_si[1] - 1), min(y1, im_si[0] - 1)
    if vx1 >= vx0 and vy1 >= vy0:
        out[vy0:vy1 + 1, vx0:vx1 + 1] = crop[vy0 - y0:vy1 - y0 + 1, vx0 - x0:vx1 - x0 + 1]
    return out


# ----------------------------------------------------------------------------- heat-maps / points
def make_gaussian(size: Tuple[int, int], sigma: float = 10, center=None) -> np.ndarray:
    """2-D Gaussian of height 1 on an ``(h, w)`` canvas centred at ``center = (x, y)``."""
    h, w = size
    x = np.arange(w, dtype=np.float32)[None, :]
    y = np.arange(h, dtype=np.float32)[:, None]
    cx, cy = (w // 2, h // 2) if center is None else center
    return np.exp(-4 * np.log(2) * ((x - cx) ** 2 + (y - cy) ** 2) / float(sigma) ** 2).astype(np.float32)


def make_gt(img: np.ndarray, labels, sigma: float = 10, one_mask_per_point: bool = False) -> np.ndarray:
    """Gaussian heat-map with one bump per point (`custom_transforms.py:246`): max-combined into
    one HxW map, or HxWxN when ``one_mask_per_point``."""
    h, w = img.shape[:2]
    if labels is None:
        return np.zeros((h, w), dtype=np.float32)
    pts = np.asarray(labels, dtype=np.float32)
    if pts.ndim == 1:
        pts = pts[None, :]
    if one_mask_per_point:
        return np.stack([make_gaussian((h, w), sigma, p) for p in pts], axis=-1)
    gt = np.zeros((h, w), dtype=np.float32)
    for p in pts:
        gt = np.maximum(gt, make_gaussian((h, w), sigma, p))
    return gt


def _extreme(mask: np.ndarray, pert: int, rng) -> np.ndarray:
    ys, xs = np.where(np.asarray(mask) > 0.5)
    if ys.size == 0:
        raise ValueError("extreme points of an empty mask")

    def pick(cond):
        idx = np.where(cond)[0]
        j = idx[len(idx) // 2] if rng is None else idx[rng.integers(0, len(idx))]
        return [xs[j], ys[j]]

    return np.array([pick(np.abs(xs - xs.min()) <= pert),   # left
                     pick(np.abs(xs - xs.max()) <= pert),   # right
                     pick(np.abs(ys - ys.min()) <= pert),   # top
                     pick(np.abs(ys - ys.max()) <= pert)],  # bottom
                    dtype=np.int64)


def extreme_points(mask: np.ndarray, pert: int = 0) -> np.ndarray:
    """4×2 ``(x, y)`` extreme points (left, right, top, bottom), each drawn at random among the
    pixels within ``pert`` of the extreme coordinate (`custom_transforms.py:21`, `:42`, `:244`)."""
    return _extreme(mask, pert, np.random.default_rng(np.random.randint(0, 2 ** 31 - 1)))


def extreme_points_fixed(mask: np.ndarray, pert: int = 0) -> np.ndarray:
    """Deterministic variant for validation: the median candidate (`custom_transforms.py:19`)."""
    return _extreme(mask, pert, None)


# ----------------------------------------------------------------------------- metrics / reports
def calc_jaccard(result: np.ndarray, gt: np.ndarray, void_pixels: Optional[np.ndarray] = None) -> float:
    """IoU of two binary masks ignoring void pixels (`train_pascal.py:291`); 1 when both empty."""
    result = np.asarray(result).astype(bool)
    gt = np.asarray(gt).astype(bool)
    if result.shape != gt.shape:
        raise ValueError("shape mismatch")
    valid = np.ones_like(gt) if void_pixels is None else ~(np.asarray(void_pixels) > 0.5)
    union = ((result | gt) & valid).sum()
    if union == 0:
        return 1.0
    return float(((result & gt) & valid).sum()) / float(union)


jaccard = calc_jaccard


def generate_param_report(logfile: str, param: Dict) -> None:
    """Dump the hyper-parameter dict as ``key: value`` lines (`train_pascal.py:169`)."""
    os.makedirs(os.path.dirname(os.path.abspath(logfile)), exist_ok=True)
    with open(logfile, "w") as fh:
        for k, v in param.items():
            fh.write(f"{k}: {v}\n")
