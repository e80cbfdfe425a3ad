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


def make_affine_params(batch: int, in_size, out_size: Tuple[int, int],
                       scales: Tuple[float, float] = (0.5, 2.0), rots: Tuple[float, float] = (0.0, 0.0),
                       flip_p: float = 0.5, rng: Optional[np.random.Generator] = None,
                       random_crop: bool = True) -> np.ndarray:
    """Per-sample *output→input* pixel maps ``[B, 6]`` = ``(a, b, c, d, e, f)`` with
    ``x_in = a·x_out + b·y_out + c`` and ``y_in = d·x_out + e·y_out + f``.

    Composition: random isotropic scale ``s`` and rotation ``θ`` about the image centre, a random
    crop window of the output size inside the scaled image (centred when the scaled image is
    smaller: the rest is padding), and a horizontal flip."""
    rng = rng or np.random.default_rng()
    per_sample = np.ndim(in_size) == 2          # [B, 2] array of (h, w): samples of different sizes (padded cache)
    if not per_sample:
        H, W = in_size
    Ho, Wo = out_size
    mats = np.zeros((batch, 6), dtype=np.float32)
    for i in range(batch):
        if per_sample:
            H, W = int(in_size[i][0]), int(in_size[i][1])
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


# --------------------------------------------------------------------------------------------------
# second-generation gather: device-resident dataset cache, bicubic, canvas clip (csrc/augment.hip)
# --------------------------------------------------------------------------------------------------
def _reference_affine2(img_u8, lab_u8, src_idx, sizes, mats, clip, out_size, interp, lab_fill, clamp, mean, std,
                       border, cout):
    """Plain-torch implementation of ``dp_affine_augment2`` (CPU path + the numerics reference of the kernel)."""
    dev = img_u8.device
    Bn = mats.shape[0]
    Ho, Wo = out_size
    ys, xs = torch.meshgrid(torch.arange(Ho, device=dev, dtype=torch.float32),
                            torch.arange(Wo, device=dev, dtype=torch.float32), indexing="ij")
    out = torch.zeros(Bn, cout, Ho, Wo, dtype=torch.float32, device=dev)
    out_lab = torch.empty(Bn, Ho, Wo, dtype=torch.uint8, device=dev) if lab_u8 is not None else None
    mean_t = torch.tensor(mean, device=dev, dtype=torch.float32).view(3, 1, 1)
    std_t = torch.tensor(std, device=dev, dtype=torch.float32).view(3, 1, 1)
    bord = torch.tensor(border, device=dev, dtype=torch.float32).view(3, 1, 1)
    for b in range(Bn):
        slot = int(src_idx[b]) if src_idx is not None else b
        H, W = (int(sizes[b][0]), int(sizes[b][1])) if sizes is not None else tuple(img_u8.shape[1:3])
        m = mats[b].to(dev)
        xi = m[0] * xs + m[1] * ys + m[2]
        yi = m[3] * xs + m[4] * ys + m[5]
        inside = torch.ones_like(xi, dtype=torch.bool)
        if clip is not None:
            q = clip[b].to(dev)
            cx, cy = q[0] * xs + q[1] * ys + q[2], q[3] * xs + q[4] * ys + q[5]
            inside = (cx > -0.5) & (cy > -0.5) & (cx < q[6] - 0.5) & (cy < q[7] - 0.5)
        src = img_u8[slot, :H, :W].permute(2, 0, 1).float() - bord           # border == 0 in this space
        grid = torch.stack([2 * xi / max(W - 1, 1) - 1, 2 * yi / max(H - 1, 1) - 1], dim=-1)[None]
        smp = F.grid_sample(src[None], grid, mode="bicubic" if interp == 2 else "bilinear", padding_mode="zeros",
                            align_corners=True)[0] + bord
        smp = torch.where(inside[None], smp, bord.expand_as(smp))
        if clamp:
            smp = smp.clamp(0.0, 255.0)
        out[b, :3] = (smp - mean_t) / std_t
        if lab_u8 is not None:
            xr, yr = torch.floor(xi + 0.5).long(), torch.floor(yi + 0.5).long()
            ok = inside & (xr >= 0) & (xr < W) & (yr >= 0) & (yr < H)
            v = lab_u8[slot][yr.clamp(0, H - 1), xr.clamp(0, W - 1)]
            out_lab[b] = torch.where(ok, v, torch.full_like(v, lab_fill))
    return out, out_lab


@torch.no_grad()
def affine_augment2(img_u8: torch.Tensor, lab_u8: Optional[torch.Tensor], mats: torch.Tensor,
                    out_size: Tuple[int, int], src_idx: Optional[torch.Tensor] = None,
                    sizes: Optional[torch.Tensor] = None, clip: Optional[torch.Tensor] = None,
                    interp: str = "bilinear", lab_fill: int = 255, clamp: bool = False,
                    mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                    border: Optional[Sequence[float]] = None, channels: int = 3,
                    out_dtype: torch.dtype = torch.bfloat16, validated: bool = False):
    """General inverse-affine gather over a (possibly padded, device-resident) uint8 sample store.

    ``img_u8`` [N,H,W,3], ``lab_u8`` [N,H,W] or None; ``mats`` [B,6] out→source maps; ``src_idx`` int32 [B] slot of
    each batch element (default ``arange``); ``sizes`` int32 [B,2] valid (h, w) of each element's slot; ``clip``
    [B,8] out→canvas map + canvas extent (pixels outside the canvas take ``border`` / ``lab_fill``);
    ``interp`` 'bilinear' | 'bicubic' (Keys a=−0.75 as ``cv2.INTER_CUBIC``); ``border``: out-of-source image value
    in 0..255 space (default: the mean colour, i.e. 0 after normalisation).  Returns
    ``(image [B,channels,Ho,Wo] channels-last, label uint8 [B,Ho,Wo] | None)``; channels ≥ 3 beyond RGB are zero
    (the guidance channel is written by ``nellipse_guidance_``)."""
    if border is None:
        border = mean
    ip = {"bilinear": 1, "bicubic": 2}[interp]
    Bn = mats.shape[0]
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 4 and img_u8.shape[-1] == 3
    assert mats.shape == (Bn, 6) and (src_idx is not None or img_u8.shape[0] == Bn)
    if B.use_hip(img_u8):
        from . import _hip_misc
        return _hip_misc.affine_augment2(img_u8, lab_u8, src_idx, sizes, mats, clip, out_size, ip, lab_fill,
                                         clamp, mean, std, border, channels, out_dtype, validated)
    out, lab = _reference_affine2(img_u8, lab_u8, src_idx, sizes, mats.float(), clip, out_size, ip, lab_fill, clamp,
                                  mean, std, border, channels)
    return out.to(out_dtype).contiguous(memory_format=torch.channels_last), lab


def make_warp_params(batch: int, sizes, rots: Tuple[float, float] = (-20.0, 20.0),
                     scales: Tuple[float, float] = (0.75, 1.25), flip_p: float = 0.5,
                     rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """Canvas→source maps [B,6] of the reference's ``RandomHorizontalFlip`` + ``ScaleNRotate`` chain
    (`/root/reference/custom_transforms.py:202-218`, `:91-126`): a flip of the source, then a rotation by ``rot``
    and scale ``sc`` about the image centre onto a canvas of the SAME size.  ``rot``/``sc`` follow the reference's
    centred-range quirk: ``(hi−lo)·U − (hi−lo)/2`` (+1 for the scale), whatever ``lo``/``hi`` are."""
    rng = rng or np.random.default_rng()
    mats = np.zeros((batch, 6), dtype=np.float32)
    for i in range(batch):
        h, w = int(sizes[i][0]), int(sizes[i][1])
        flip = rng.random() < flip_p
        rot = (rots[1] - rots[0]) * rng.random() - (rots[1] - rots[0]) / 2
        sc = (scales[1] - scales[0]) * rng.random() - (scales[1] - scales[0]) / 2 + 1
        # forward (source→canvas) matrix of cv2.getRotationMatrix2D((w/2, h/2), rot, sc); invert it
        a = sc * math.cos(math.radians(rot))
        bb = sc * math.sin(math.radians(rot))
        cx, cy = w / 2.0, h / 2.0
        fwd = np.array([[a, bb, (1 - a) * cx - bb * cy], [-bb, a, bb * cx + (1 - a) * cy], [0, 0, 1]], dtype=np.float64)
        inv = np.linalg.inv(fwd)
        if flip:    # the flip acts on the source first: x_src = (w − 1) − x
            inv = np.array([[-1, 0, w - 1], [0, 1, 0], [0, 0, 1]], dtype=np.float64) @ inv
        mats[i] = inv[:2].reshape(-1)
    return mats


@torch.no_grad()
def check_slots(src_idx_host, sizes_host, nslots: int, slot_hw) -> None:
    """Host-side validation of slot indices / valid sizes (NumPy copies) — lets the device path skip its own
    checks, which would read device tensors and synchronise the host with the GPU every step."""
    if src_idx_host is not None and len(src_idx_host) and (min(src_idx_host) < 0 or max(src_idx_host) >= nslots):
        raise IndexError(f"src_idx outside the store ({nslots} slots)")
    if sizes_host is not None and len(sizes_host):
        sz = np.asarray(sizes_host)
        if sz.min() < 1 or sz[:, 0].max() > slot_hw[0] or sz[:, 1].max() > slot_hw[1]:
            raise ValueError("sizes outside the slot dimensions")


def mask_bbox(mask_u8: torch.Tensor, mats: torch.Tensor, src_idx=None, sizes=None, validated: bool = False) -> torch.Tensor:
    """int32 [B,4] inclusive (x_min, y_min, x_max, y_max) of the nearest-warped mask on its canvas; (·,·,−1,−1) when
    empty — ``helpers.get_bbox`` of ``ScaleNRotate``'s output (`/root/reference/custom_transforms.py:359`)."""
    if B.use_hip(mask_u8):
        from . import _hip_misc
        return _hip_misc.mask_bbox(mask_u8, src_idx, sizes, mats, validated)
    Bn = mats.shape[0]
    out = torch.empty(Bn, 4, dtype=torch.int32)
    for b in range(Bn):
        slot = int(src_idx[b]) if src_idx is not None else b
        H, W = (int(sizes[b][0]), int(sizes[b][1])) if sizes is not None else tuple(mask_u8.shape[1:3])
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                                indexing="ij")
        m = mats[b].float()
        xr = torch.floor(m[0] * xs + m[1] * ys + m[2] + 0.5).long()
        yr = torch.floor(m[3] * xs + m[4] * ys + m[5] + 0.5).long()
        ok = (xr >= 0) & (xr < W) & (yr >= 0) & (yr < H)
        on = ok & (mask_u8[slot][yr.clamp(0, H - 1), xr.clamp(0, W - 1)] != 0)
        if on.any():
            yy, xx = torch.where(on)
            out[b] = torch.tensor([xx.min(), yy.min(), xx.max(), yy.max()], dtype=torch.int32)
        else:
            out[b] = torch.tensor([2 ** 31 - 1, 2 ** 31 - 1, -1, -1], dtype=torch.int32)
    return out


@torch.no_grad()
def compose_crop(mats: torch.Tensor, bbox: torch.Tensor, out_size: Tuple[int, int], relax: int = 50,
                 zero_pad: bool = True, sizes=None, slot_size: Tuple[int, int] = (0, 0)):
    """Maps of ``CropFromMaskStatic(relax, zero_pad)`` + ``FixedResize(out_size)`` composed with the warp
    (`/root/reference/custom_transforms.py:343-371`, `:156-196`).  Returns ``(mats [B,6] out→source,
    clip [B,8] out→canvas + extent, crop_bbox int32 [B,4])``."""
    Ho, Wo = out_size
    if B.use_hip(mats):
        from . import _hip_misc
        return _hip_misc.compose_crop(mats, bbox, sizes, slot_size, out_size, relax, zero_pad)
    Bn = mats.shape[0]
    om, oc, ob = torch.zeros(Bn, 6), torch.zeros(Bn, 8), torch.zeros(Bn, 4, dtype=torch.int32)
    for b in range(Bn):
        H, W = (int(sizes[b][0]), int(sizes[b][1])) if sizes is not None else slot_size
        x0, y0, x1, y1 = [int(v) for v in bbox[b]]
        if x1 < 0:
            om[b] = torch.tensor([0, 0, -1e6, 0, 0, -1e6])
            oc[b] = torch.tensor([0, 0, -1e6, 0, 0, -1e6, W, H])
            ob[b] = torch.tensor([0, 0, -1, -1], dtype=torch.int32)
            continue
        x0, y0, x1, y1 = x0 - relax, y0 - relax, x1 + relax, y1 + relax
        if not zero_pad:
            x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, W - 1), min(y1, H - 1)
        ob[b] = torch.tensor([x0, y0, x1, y1], dtype=torch.int32)
        sx, sy = (x1 - x0 + 1) / Wo, (y1 - y0 + 1) / Ho
        ca, cc, ce, cf = sx, 0.5 * sx - 0.5 + x0, sy, 0.5 * sy - 0.5 + y0
        oc[b] = torch.tensor([ca, 0, cc, 0, ce, cf, W, H])
        m = mats[b].float()
        om[b] = torch.stack([m[0] * ca, m[1] * ce, m[0] * cc + m[1] * cf + m[2],
                             m[3] * ca, m[4] * ce, m[3] * cc + m[4] * cf + m[5]])
    return om, oc, ob


_IDENT: dict = {}


def _identity_mats(n: int, device) -> torch.Tensor:
    """[n, 6] identity affine rows ON ``device``, built once per (n, device): a host tensor moved with ``.to(device)``
    is a blocking copy from pageable memory — it waits for everything queued on the stream, i.e. one full host-device
    synchronisation per batch in the instance task's loader (found as 100 ms of "data wait" per step behind a loader
    whose own host cost is 1 ms, tools/loader_host_cost.py)."""
    key = (int(n), str(device))
    t = _IDENT.get(key)
    if t is None:
        t = torch.zeros(n, 6, dtype=torch.float32, device=device)
        t[:, 0] = 1.0
        t[:, 4] = 1.0
        _IDENT[key] = t
    return t


@torch.no_grad()
def extreme_points(mask_u8: torch.Tensor, random: bool = True, seed: int = 0) -> torch.Tensor:
    """int32 [B,4,2] (x, y) of the left / right / top / bottom extreme pixel of each mask [B,H,W]; among the mask
    pixels on the extreme line one is drawn at random (training) or the median one taken (validation) —
    ``helpers.extreme_points`` / ``extreme_points_fixed`` with ``pert=0`` (`/root/reference/custom_transforms.py:40-42`).
    Empty masks give −1."""
    if B.use_hip(mask_u8):
        from . import _hip_misc
        bbox = _hip_misc.mask_bbox(mask_u8, None, None, _identity_mats(mask_u8.shape[0], mask_u8.device))
        return _hip_misc.extreme_points(mask_u8, bbox, random, seed)
    Bn, H, W = mask_u8.shape
    pts = torch.full((Bn, 4, 2), -1, dtype=torch.int32)
    g = torch.Generator().manual_seed(seed)
    for b in range(Bn):
        yy, xx = torch.where(mask_u8[b] != 0)
        if yy.numel() == 0:
            continue
        for side, (fixed_is_x, val) in enumerate(((True, xx.min()), (True, xx.max()), (False, yy.min()), (False, yy.max()))):
            cand = torch.where(xx == val)[0] if fixed_is_x else torch.where(yy == val)[0]
            other = (yy if fixed_is_x else xx)[cand]
            other, _ = torch.sort(other)
            k = int(torch.randint(0, len(other), (1,), generator=g)) if random else len(other) // 2
            pts[b, side] = torch.tensor([int(val), int(other[k])] if fixed_is_x else [int(other[k]), int(val)],
                                        dtype=torch.int32)
    return pts


@torch.no_grad()
def nellipse_guidance_(out: torch.Tensor, pts: torch.Tensor, channel: int = 3, alpha: float = 0.6,
                       sigma: float = 10.0, mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    """Writes the ``NEllipseWithGaussians`` guidance map (0..255, then ``(v−mean)/std``) of the four points
    ``pts`` [B,4,2] into channel ``channel`` of the channels-last tensor ``out`` [B,C,H,W]
    (`/root/reference/custom_transforms.py:30-51`; maths of data/nellipse.py)."""
    if B.use_hip(out):
        from . import _hip_misc
        return _hip_misc.nellipse_guidance(out, pts, channel, alpha, sigma, mean, std)
    Bn, C, H, W = out.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    for b in range(Bn):
        f = pts[b].float()
        if (f < 0).any():
            out[b, channel] = (0.0 - mean) / std
            continue
        d = sum(torch.sqrt((xs - f[k, 0]) ** 2 + (ys - f[k, 1]) ** 2) for k in range(4))
        dref = max(float(sum(torch.sqrt(((f - f[k]) ** 2).sum(1)))) for k in range(4))
        dmin = float(d.min())
        z1 = (d <= dref).float() if dref - dmin < 1e-6 else (1 - (d - dmin) / (dref - dmin)).clamp(0, 1)
        z2 = torch.zeros_like(z1)
        for k in range(4):
            z2 = torch.maximum(z2, torch.exp(-4 * math.log(2) * ((xs - f[k, 0]) ** 2 + (ys - f[k, 1]) ** 2) / sigma ** 2))
        z = 255 * z1 + alpha * 255 * z2
        z = (z * (255.0 / max(float(z.max()), 1e-10))).clamp(max=255.0)
        out[b, channel] = ((z - mean) / std).to(out.dtype)
    return out


@torch.no_grad()
# ---- batched device forms of the transforms the reference's script does not call ------------------------------------
# (`/root/reference/custom_transforms.py`: CreateBBMask :67-74, ExtremePoints :221-251, AddConfidenceMap :253-298,
# CropFromMask :377-452.)  Tensor expressions over the whole batch on whatever device the masks live on — no per-sample
# host loop, no host read; the extreme points and bounding boxes come from the kernels above.  Values follow
# data/transforms.py (the CPU forms, which the tests compare against), quirks included.
def _grid(h: int, w: int, device):
    ys = torch.arange(h, device=device, dtype=torch.float32).view(1, h, 1)
    xs = torch.arange(w, device=device, dtype=torch.float32).view(1, 1, w)
    return ys, xs


def _identity_bbox(mask_u8: torch.Tensor) -> torch.Tensor:
    return mask_bbox(mask_u8, _identity_mats(mask_u8.shape[0], mask_u8.device), validated=True)


@torch.no_grad()
def bbox_mask(mask_u8: torch.Tensor) -> torch.Tensor:
    """``CreateBBMask``: float32 [B,H,W], 255 outside the mask's bounding box and 0 inside (the box's last row and
    column stay 255, as in the CPU form's half-open slices); all 255 for an empty mask."""
    bb = _identity_bbox(mask_u8).float()                                   # x0, y0, x1, y1 (x1 < 0: empty)
    ys, xs = _grid(mask_u8.shape[1], mask_u8.shape[2], mask_u8.device)
    x0, y0, x1, y1 = (bb[:, i].view(-1, 1, 1) for i in range(4))
    inside = (xs >= x0) & (xs < x1) & (ys >= y0) & (ys < y1)
    return torch.where(inside, 0.0, 255.0)


@torch.no_grad()
def extreme_points_heatmap(mask_u8: torch.Tensor, sigma: float = 10.0, random: bool = False, seed: int = 0) -> torch.Tensor:
    """``ExtremePoints`` (pert = 0): float32 [B,H,W], the maximum of four Gaussians
    ``exp(-4 ln2 ((x-cx)^2 + (y-cy)^2) / sigma^2)`` at the left / right / top / bottom extreme pixel
    (``helpers.make_gt``); zeros for an empty mask."""
    pts = extreme_points(mask_u8, random, seed).to(mask_u8.device).float()  # [B,4,2] (x, y), -1 when empty
    ys, xs = _grid(mask_u8.shape[1], mask_u8.shape[2], mask_u8.device)
    cx, cy = pts[:, :, 0].view(-1, 4, 1, 1), pts[:, :, 1].view(-1, 4, 1, 1)
    d2 = (xs.unsqueeze(1) - cx) ** 2 + (ys.unsqueeze(1) - cy) ** 2
    heat = torch.exp(-4.0 * math.log(2.0) * d2 / float(sigma) ** 2).amax(dim=1)
    empty = (pts[:, 0, 0] < 0).view(-1, 1, 1)
    return torch.where(empty, torch.zeros_like(heat), heat)


@torch.no_grad()
def confidence_map(mask_u8: torch.Tensor, hm_type: str = "l1l2", tau: float = 1.0, random: bool = False,
                   seed: int = 0) -> torch.Tensor:
    """``AddConfidenceMap``'s extra channel: float32 [B,H,W] in 0..255, min-max normalised per sample.
    ``l1l2``: exp(-(l1 + l2) / 2 tau) of the distances to the chords left-right and top-bottom of the extreme points,
    each in units of the OTHER chord's half length (``generate_mvL1L2_image_skewed_axes``); ``gaussian``: the
    multivariate Gaussian fitted to the mask pixels, covariance / 0.5 (``generate_mvgauss_image``; the CPU form
    ignores ``tau`` for this type, so does this one).  Zeros when the mask is empty or full."""
    bsz, h, w = mask_u8.shape
    dev = mask_u8.device
    m = (mask_u8 != 0)
    ys, xs = _grid(h, w, dev)
    cnt = m.flatten(1).sum(1).float()
    if hm_type == "l1l2":
        pts = extreme_points(mask_u8, random, seed).to(dev).float()

        def dist(p, q):                                  # |cross(q - p, x - p)| / |q - p|, and |q - p|
            d = q - p
            n = torch.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2) + 1e-6
            num = (d[:, 1].view(-1, 1, 1) * (xs - p[:, 0].view(-1, 1, 1))
                   - d[:, 0].view(-1, 1, 1) * (ys - p[:, 1].view(-1, 1, 1))).abs()
            return num / n.view(-1, 1, 1), n
        d_lr, len_lr = dist(pts[:, 0], pts[:, 1])
        d_tb, len_tb = dist(pts[:, 2], pts[:, 3])
        d1 = d_lr / (0.5 * len_tb + 1e-6).view(-1, 1, 1)
        d2 = d_tb / (0.5 * len_lr + 1e-6).view(-1, 1, 1)
        hmap = torch.exp(-0.5 * (d1 + d2 + torch.sqrt(d1 ** 2 + d2 ** 2)) / max(float(tau), 1e-6))
    elif hm_type == "gaussian":
        mf = m.float()
        n = cnt.clamp_min(1.0).view(-1, 1, 1)
        mx, my = (mf * xs).sum((1, 2), keepdim=True) / n, (mf * ys).sum((1, 2), keepdim=True) / n
        dx, dy = xs - mx, ys - my
        nm1 = (cnt - 1.0).clamp_min(1.0).view(-1, 1, 1)                   # np.cov: unbiased
        sxx = (mf * dx * dx).sum((1, 2), keepdim=True) / nm1 / 0.5 + 1e-3
        syy = (mf * dy * dy).sum((1, 2), keepdim=True) / nm1 / 0.5 + 1e-3
        sxy = (mf * dx * dy).sum((1, 2), keepdim=True) / nm1 / 0.5
        det = sxx * syy - sxy * sxy
        hmap = torch.exp(-0.5 * (syy * dx * dx - 2 * sxy * dx * dy + sxx * dy * dy) / det)
        hmap = torch.where((cnt < 2).view(-1, 1, 1), torch.zeros_like(hmap), hmap)
    else:
        raise ValueError(f"unknown hm_type {hm_type}")
    lo, hi = hmap.amin((1, 2), keepdim=True), hmap.amax((1, 2), keepdim=True)
    out = (hmap - lo) / (hi - lo + 1e-10) * 255.0
    trivial = ((cnt == 0) | (cnt == h * w)).view(-1, 1, 1)
    return torch.where(trivial, torch.zeros_like(out), out)


@torch.no_grad()
def dynamic_crop_relax(bbox: torch.Tensor, d: int = 512, dz: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``CropFromMask``'s relax per sample (int32 [B]) from the inclusive mask boxes ``bbox`` [B,4]: the border that
    makes the object take ``dz`` of the ``d`` x ``d`` input — ``ceil(max((d - side zoom) / (2 zoom), thresh))`` with
    ``zoom = dz / side``.  ``dz``: [B] (training: drawn per call in [floor(sqrt(.45 d^2)), floor(sqrt(.6 d^2)))) or
    None = validation's ``floor(sqrt(d^2 / 2))``.  Feed the result to ``expand_bbox`` + ``compose_crop(relax=0)``."""
    dz0 = float(int(math.sqrt(d ** 2 * 0.5)))
    thresh = ((d - dz0) * (d / 5)) / (2 * dz0)
    bb = bbox.float()
    side = torch.maximum(torch.maximum(bb[:, 2] - bb[:, 0], bb[:, 3] - bb[:, 1]), torch.ones_like(bb[:, 0]))
    dzv = torch.full_like(side, dz0) if dz is None else dz.to(side.device).float()
    zoom = dzv / side
    relax = torch.clamp_min((d - side * zoom) / (2 * zoom), thresh)
    return torch.ceil(relax).to(torch.int32)


def expand_bbox(bbox: torch.Tensor, relax: torch.Tensor) -> torch.Tensor:
    """Inclusive boxes grown by a per-sample border (empty boxes, x1 < x0, stay empty)."""
    r = relax.to(bbox.device, torch.int32)
    out = torch.stack([bbox[:, 0] - r, bbox[:, 1] - r, bbox[:, 2] + r, bbox[:, 3] + r], dim=1).to(torch.int32)
    return torch.where((bbox[:, 2] < bbox[:, 0]).view(-1, 1), bbox.to(torch.int32), out)


@torch.no_grad()
def paste_jaccard(logits: torch.Tensor, crop_bbox: torch.Tensor, gt_u8: torch.Tensor, void_u8: Optional[torch.Tensor],
                  src_idx, sizes, relax: int = 50, thresholds=(0.3, 0.5, 0.8), validated: bool = False) -> torch.Tensor:
    """Validation metric of the instance task without leaving the device
    (`/root/reference/train_pascal.py:280-306`): for every object of the batch the crop-sized ``sigmoid(logits)``
    is pasted back into its full-size canvas (``helpers.crop2fullmask``: bicubic resize to ``crop_bbox``, ``relax``
    border suppressed, parts outside the image dropped), thresholded, and compared with the full-size object mask
    ``gt_u8[src_idx]`` where ``void_u8`` is 0.  Returns int32 ``[B, T, 2]``: intersection and union pixel counts;
    ``helpers.calc_jaccard`` = inter / union (1 when union is 0).  ``logits``: [B,1,Hc,Wc]."""
    if B.use_hip(logits):
        from . import _hip_misc
        lg = logits if logits.dtype in (torch.float32, torch.bfloat16) else logits.float()
        return _hip_misc.paste_jaccard(lg.contiguous(), crop_bbox, gt_u8, void_u8, src_idx, sizes, relax, thresholds,
                                       validated=validated)
    import numpy as np
    from ..data import helpers
    bsz = logits.shape[0]
    out = torch.zeros(bsz, len(thresholds), 2, dtype=torch.int32)
    prob = torch.sigmoid(logits.float()).reshape(bsz, logits.shape[-2], logits.shape[-1]).cpu().numpy()
    for b in range(bsz):
        slot = int(src_idx[b]) if src_idx is not None else b
        h, w = int(sizes[b][0]), int(sizes[b][1])
        gt = gt_u8[slot, :h, :w].cpu().numpy() != 0
        valid = np.ones_like(gt) if void_u8 is None else void_u8[slot, :h, :w].cpu().numpy() == 0
        bb = [int(v) for v in crop_bbox[b]]
        full = helpers.crop2fullmask(prob[b], bb, im_size=(h, w), relax=relax)
        for t, th in enumerate(thresholds):
            pr = full > th
            out[b, t, 0] = int(((pr & gt) & valid).sum())
            out[b, t, 1] = int(((pr | gt) & valid).sum())
    return out


def instance_augment(img_u8: torch.Tensor, gt_u8: torch.Tensor, src_idx: torch.Tensor, sizes: torch.Tensor,
                     warp_mats: torch.Tensor, out_size: Tuple[int, int] = (512, 512), relax: int = 50,
                     zero_pad: bool = True, alpha: float = 0.6, random_points: bool = True, seed: int = 0,
                     out_dtype: torch.dtype = torch.bfloat16, validated: bool = False):
    """The reference's whole training transform chain for the instance task
    (`/root/reference/train_pascal.py:123-134`) on the device: Flip → ScaleNRotate (``warp_mats``) →
    CropFromMaskStatic(relax, zero_pad) → FixedResize(out_size) → NEllipseWithGaussians(alpha) →
    ConcatInputs → ToTensor.  Returns ``(concat [B,4,Ho,Wo] channels-last, values 0..255 as the reference feeds
    them; crop_gt [B,1,Ho,Wo] float {0,1}; crop_bbox int32 [B,4])``.  ONE bicubic resampling replaces the
    reference's two (warp, then resize): fewer interpolation passes, same geometry."""
    warp_mats = warp_mats.to(img_u8.device)
    bbox = mask_bbox(gt_u8, warp_mats, src_idx, sizes, validated)
    mats, clip, crop_bbox = compose_crop(warp_mats, bbox, out_size, relax, zero_pad, sizes, tuple(img_u8.shape[1:3]))
    x, crop_gt = affine_augment2(img_u8, gt_u8, mats, out_size, src_idx, sizes, clip, interp="bicubic", lab_fill=0,
                                 clamp=True, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), border=(0.0, 0.0, 0.0),
                                 channels=4, out_dtype=out_dtype, validated=validated)
    pts = extreme_points(crop_gt, random_points, seed)
    nellipse_guidance_(x, pts, 3, alpha)
    return x, crop_gt.unsqueeze(1).to(torch.float32), crop_bbox
