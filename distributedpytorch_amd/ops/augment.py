"""GPU-side augmentation, SURVEY.md §2.5 K8/K9.

The reference augments on the CPU inside DataLoader workers: ``cv2.flip``
(`/root/reference/custom_transforms.py:212`), ``cv2.warpAffine`` scale+rotate (`:116-126`), mask
crops and resizes (`:186-193`, `:359-366`), then ``ToTensor`` (`:497-498`).  On MI355X the whole
chain is ONE inverse-affine gather per batch: each output pixel maps through a per-sample 2×3
matrix (scale, rotation, crop offset and horizontal flip already composed), the image is sampled
bilinearly (bicubic optional) with out-of-bounds taps equal to the mean colour, the label is
sampled nearest with out-of-bounds = 255 (ignore), and the epilogue writes normalised NHWC bf16
(K9: HWC→CHW, cast and ``(x-mean)/std`` folded in).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _backend as B

IMAGENET_MEAN = (0.485 * 255, 0.456 * 255, 0.406 * 255)
IMAGENET_STD = (0.229 * 255, 0.224 * 255, 0.225 * 255)


def make_affine_params(batch: int, in_size: Tuple[int, int], out_size: Tuple[int, int],
                       scales: Tuple[float, float] = (0.5, 2.0), rots: Tuple[float, float] = (0.0, 0.0),
                       flip_p: float = 0.5, rng: Optional[np.random.Generator] = None,
                       random_crop: bool = True) -> np.ndarray:
    """Per-sample *output→input* pixel maps ``[B, 6]`` = ``(a, b, c, d, e, f)`` with
    ``x_in = a·x_out + b·y_out + c`` and ``y_in = d·x_out + e·y_out + f``.

    Composition: random isotropic scale ``s`` and rotation ``θ`` about the image centre, a random
    crop window of the output size inside the scaled image (centred when the scaled image is
    smaller: the rest is padding), and a horizontal flip."""
    rng = rng or np.random.default_rng()
    H, W = in_size
    Ho, Wo = out_size
    mats = np.zeros((batch, 6), dtype=np.float32)
    for i in range(batch):
        s = rng.uniform(scales[0], scales[1])
        th = math.radians(rng.uniform(rots[0], rots[1]))
        flip = rng.random() < flip_p
        sh, sw = H * s, W * s  # scaled image extent
        if random_crop:
            ox = rng.uniform(0, sw - Wo) if sw > Wo else (sw - Wo) / 2.0
            oy = rng.uniform(0, sh - Ho) if sh > Ho else (sh - Ho) / 2.0
        else:
            ox, oy = (sw - Wo) / 2.0, (sh - Ho) / 2.0
        cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
        cos, sin = math.cos(th) / s, math.sin(th) / s
        # output pixel -> scaled-image coords (+ox,+oy) -> centre -> inverse rotate/scale -> input
        # u = x_out + ox - (sw-1)/2 ; v = y_out + oy - (sh-1)/2
        ux, uy = ox - (sw - 1) / 2.0, oy - (sh - 1) / 2.0
        sx = -1.0 if flip else 1.0
        if flip:  # mirror output x about the crop centre
            ux = (Wo - 1) + ox - (sw - 1) / 2.0
        a, b = cos * sx, sin
        d, e = -sin * sx, cos
        c = cos * ux + sin * uy + cx
        f = -sin * ux + cos * uy + cy
        mats[i] = (a, b, c, d, e, f)
    return mats


def _reference_affine(img_u8, lab_u8, mats, out_size, mean, std):
    Bn, H, W, _ = img_u8.shape
    Ho, Wo = out_size
    dev = img_u8.device
    ys, xs = torch.meshgrid(torch.arange(Ho, device=dev, dtype=torch.float32),
                            torch.arange(Wo, device=dev, dtype=torch.float32), indexing="ij")
    m = mats.to(dev).view(Bn, 6, 1, 1)
    xi = m[:, 0] * xs + m[:, 1] * ys + m[:, 2]
    yi = m[:, 3] * xs + m[:, 4] * ys + m[:, 5]
    grid = torch.stack([2 * xi / max(W - 1, 1) - 1, 2 * yi / max(H - 1, 1) - 1], dim=-1)
    mean_t = torch.tensor(mean, device=dev, dtype=torch.float32).view(1, 3, 1, 1)
    std_t = torch.tensor(std, device=dev, dtype=torch.float32).view(1, 3, 1, 1)
    x = (img_u8.permute(0, 3, 1, 2).float() - mean_t) / std_t
    out = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    lab = None
    if lab_u8 is not None:
        # nearest with ignore fill: sample (label - 255) so zero padding means "255"
        l = lab_u8.view(Bn, 1, H, W).float() - 255.0
        xr, yr = torch.floor(xi + 0.5), torch.floor(yi + 0.5)
        gridn = torch.stack([2 * xr / max(W - 1, 1) - 1, 2 * yr / max(H - 1, 1) - 1], dim=-1)
        lab = (F.grid_sample(l, gridn, mode="nearest", padding_mode="zeros", align_corners=True)
               + 255.0).round().clamp(0, 255).to(torch.uint8).view(Bn, Ho, Wo)
    return out, lab


@torch.no_grad()
def affine_augment(img_u8: torch.Tensor, lab_u8: Optional[torch.Tensor], mats, out_size: Tuple[int, int],
                   mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                   out_dtype: torch.dtype = torch.bfloat16):
    """``img_u8`` [B,H,W,3] uint8, ``lab_u8`` [B,H,W] uint8 (or None), ``mats`` [B,6] fp32.

    Returns ``(image [B,3,Ho,Wo] channels-last out_dtype, label [B,Ho,Wo] uint8)``."""
    if not torch.is_tensor(mats):
        mats = torch.from_numpy(np.asarray(mats, dtype=np.float32))
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 4 and img_u8.shape[-1] == 3
    assert mats.shape == (img_u8.shape[0], 6)
    if B.use_hip(img_u8):
        from . import _hip_misc
        return _hip_misc.affine_augment(img_u8, lab_u8, mats, out_size, mean, std, out_dtype)
    out, lab = _reference_affine(img_u8, lab_u8, mats.float(), out_size, mean, std)
    return out.to(out_dtype).contiguous(memory_format=torch.channels_last), lab
