"""ctypes bindings for csrc/loss_upsample.hip and csrc/misc.hip with host-side validation."""
from __future__ import annotations

from ctypes import c_float, c_int, c_long, c_void_p
from typing import Optional, Tuple

import torch

from . import _backend as B
from ._hip_bn import as_nhwc

_sigs_done = False


def _lib():
    global _sigs_done
    lib = B.lib()
    if not _sigs_done:
        V, I, L, F = c_void_p, c_int, c_long, c_float
        lib.dp_softmax_ce_fwd.argtypes = [V, V, I, L, I, I, V, I, V]
        lib.dp_softmax_ce_bwd.argtypes = [V, V, I, L, I, I, V, V, V, V]
        lib.dp_upsample_ce_fwd.argtypes = [V, V, I, I, I, I, I, I, I, I, I, V, I, V]
        lib.dp_upsample_ce_bwd.argtypes = [V, V, I, I, I, I, I, I, I, I, I, I, V, V, V, V]
        lib.dp_bilinear_fwd.argtypes = [V, V, I, I, I, I, I, I, V]
        lib.dp_bilinear_bwd.argtypes = [V, V, I, I, I, I, I, I, V]
        lib.dp_sgd_flat.argtypes = [V, V, V, V, L, F, F, F, F, I, I, V]
        lib.dp_maxpool3x3s2_fwd.argtypes = [V, V, V, I, I, I, I, I, I, V]
        lib.dp_maxpool3x3s2_bwd.argtypes = [V, V, V, I, I, I, I, I, I, V]
        lib.dp_gap_fwd.argtypes = [V, V, I, I, I, V]
        lib.dp_gap_bwd.argtypes = [V, V, I, I, I, V]
        lib.dp_affine_augment.argtypes = [V, V, V, V, V, I, I, I, I, I, I, V, V, V]
        lib.dp_confusion.argtypes = [V, I, V, I, L, I, I, I, V, V]
        lib.dp_pad_channels.argtypes = [V, V, L, I, I, V]
        lib.dp_bilinear_fwd_ld.argtypes = [V, V, I, I, I, I, I, I, I, V]
        lib.dp_bilinear_bwd_ld.argtypes = [V, V, I, I, I, I, I, I, I, V]
        lib.dp_stem_pack.argtypes = [V, V, L, I, I, I, I, V]
        lib.dp_affine_augment2.argtypes = [V, V, V, V, V, V, V, V] + [I] * 9 + [V, V, V, V]
        lib.dp_mask_bbox.argtypes = [V, V, V, V, V, I, I, I, V]
        lib.dp_bce_blocks.argtypes = [L]
        lib.dp_balanced_bce_fwd.argtypes = [V, V, V, I, I, V, L, I, F, F, F, V, V, V]
        lib.dp_balanced_bce_bwd.argtypes = [V, V, V, I, I, V, L, V, V, V, V, V, V]
        lib.dp_paste_jaccard.argtypes = [V, I, V, V, V, V, V, V, V] + [I] * 7 + [V]
        lib.dp_compose_crop.argtypes = [V, V, V, V, V, V] + [I] * 7 + [V]
        lib.dp_extreme_points.argtypes = [V, V, V, I, I, I, I, c_int, V]
        lib.dp_nellipse_guidance.argtypes = [V, V, V, I, I, I, I, I, F, F, F, F, V]
        _sigs_done = True
    return lib


def _req_bf16_nhwc(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name}: HIP path needs bf16, got {t.dtype}")
    if t.dim() != 4:
        raise ValueError(f"{name}: expected [N,C,H,W]")
    t = as_nhwc(t)
    if t.data_ptr() % 16:
        raise ValueError(f"{name}: data pointer must be 16-byte aligned")
    return t


_LABEL_KIND = {torch.uint8: 0, torch.int32: 1, torch.int64: 2}


def _label(target: torch.Tensor, shape) -> Tuple[torch.Tensor, int]:
    if target.dtype not in _LABEL_KIND:
        target = target.long()
    if tuple(target.shape) != tuple(shape):
        raise ValueError(f"target shape {tuple(target.shape)} != {tuple(shape)}")
    return target.contiguous(), _LABEL_KIND[target.dtype]


# ------------------------------------------------------------------ softmax CE
_LOSS_REPS = 2048      # >= the largest grid of the loss kernels (csrc/loss_upsample.hip, csrc/upsample_ce.hip)


def softmax_ce_fwd(logits: torch.Tensor, target: torch.Tensor, ignore_index: int):
    logits = _req_bf16_nhwc(logits, "logits")
    n, c, h, w = logits.shape
    target, kind = _label(target, (n, h, w))
    reps = _LOSS_REPS if B.deterministic() else 1         # one (loss, count) pair per block, summed in order
    sums = torch.zeros(reps, 2, dtype=torch.float32, device=logits.device)
    B.check(_lib().dp_softmax_ce_fwd(B.ptr(logits), B.ptr(target), kind, n * h * w, c, int(ignore_index),
                                     B.ptr(sums), reps, B.stream_ptr()), "softmax_ce_fwd")
    sums = sums.sum(0) if reps > 1 else sums[0]
    loss = sums[0] / sums[1].clamp_min(1.0)
    return loss, sums, logits, target


def softmax_ce_bwd(logits, target, ignore_index, sums, gloss):
    n, c, h, w = logits.shape
    kind = _LABEL_KIND[target.dtype]
    g = gloss.detach().reshape(1).float().contiguous()
    dlogits = torch.empty_like(logits)
    B.check(_lib().dp_softmax_ce_bwd(B.ptr(logits), B.ptr(target), kind, n * h * w, c, int(ignore_index),
                                     B.ptr(sums), B.ptr(g), B.ptr(dlogits), B.stream_ptr()), "softmax_ce_bwd")
    return dlogits


# ------------------------------------------------------------------ fused upsample + softmax CE
def _strided_nhwc(t: torch.Tensor):
    """(tensor, ld): ``t`` [N,C,H,W] bf16 whose memory is NHWC with a channel stride ld >= C — a
    channels-last tensor (ld = C) or the ``[:, :C]`` view of one (the classifier's padded output)."""
    if t.dtype != torch.bfloat16 or t.dim() != 4:
        raise TypeError("upsample_ce: logits must be bf16 [N,C,H,W]")
    n, c, h, w = t.shape
    ld = t.stride(3)
    if not (t.stride(1) == 1 and ld >= c and t.stride(2) == w * ld and t.stride(0) == h * w * ld):
        t = as_nhwc(t)
        ld = c
    return t, ld


def upsample_ce_fwd(logits: torch.Tensor, target: torch.Tensor, ignore_index: int):
    """Σloss / #valid of CE(bilinear_align_corners(logits → target size), target) without
    materialising the upsampled logits.  Returns None when the shape is not taken (caller falls back)."""
    x, ld = _strided_nhwc(logits)
    n, c, hi, wi = x.shape
    ho, wo = target.shape[1:]
    target, kind = _label(target, (n, ho, wo))
    reps = _LOSS_REPS if B.deterministic() else 1
    sums = torch.zeros(reps, 2, dtype=torch.float32, device=x.device)
    st = _lib().dp_upsample_ce_fwd(B.ptr(x), B.ptr(target), kind, n, c, ld, hi, wi, ho, wo, int(ignore_index),
                                   B.ptr(sums), reps, B.stream_ptr())
    if st == -2:
        return None
    B.check(st, "upsample_ce_fwd")
    sums = sums.sum(0) if reps > 1 else sums[0]
    return sums[0] / sums[1].clamp_min(1.0), sums, x, target, ld


def upsample_ce_bwd(x, ld, target, ignore_index, sums, gloss):
    n, c, hi, wi = x.shape
    ho, wo = target.shape[1:]
    kind = _LABEL_KIND[target.dtype]
    g = gloss.detach().reshape(1).float().contiguous()
    ldo = (c + 7) // 8 * 8
    dx = torch.empty((n, ldo, hi, wi), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    B.check(_lib().dp_upsample_ce_bwd(B.ptr(x), B.ptr(target), kind, n, c, ld, ldo, hi, wi, ho, wo,
                                      int(ignore_index), B.ptr(sums), B.ptr(g), B.ptr(dx), B.stream_ptr()),
            "upsample_ce_bwd")
    return dx[:, :c] if ldo != c else dx


# ------------------------------------------------------------------ bilinear
def bilinear_fwd(x: torch.Tensor, size: Tuple[int, int], out=None) -> torch.Tensor:
    x = _req_bf16_nhwc(x, "x")
    n, c, hi, wi = x.shape
    ho, wo = size
    ldo = c
    if out is not None:
        buf, off, cs = out                       # ops.concat slot: write the channel slice in place
        if (cs != c or buf.dtype != x.dtype or tuple(buf.shape[2:]) != (ho, wo) or buf.shape[0] != n
                or not buf.is_contiguous(memory_format=torch.channels_last) or off % 8 or c % 8 or buf.shape[1] % 8):
            raise ValueError("bilinear_fwd: concat slot does not match the output")
        y, ldo = buf[:, off:off + c], buf.shape[1]
    else:
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    B.check(_lib().dp_bilinear_fwd_ld(B.ptr(x), B.ptr(y), n, c, hi, wi, ho, wo, ldo, B.stream_ptr()), "bilinear_fwd")
    return y


def bilinear_bwd(dy: torch.Tensor, in_size: Tuple[int, int]) -> torch.Tensor:
    from .concat import row_stride
    ld = row_stride(dy) if dy.dtype == torch.bfloat16 else None
    if ld is None or ld % 8 or dy.data_ptr() % 16 or dy.shape[1] % 8:
        dy = _req_bf16_nhwc(dy, "dy")
        ld = dy.shape[1]
    n, c, ho, wo = dy.shape
    hi, wi = in_size
    dx = torch.empty((n, c, hi, wi), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    B.check(_lib().dp_bilinear_bwd_ld(B.ptr(dy), B.ptr(dx), n, c, hi, wi, ho, wo, ld, B.stream_ptr()), "bilinear_bwd")
    return dx


# ------------------------------------------------------------------ pooling
def maxpool3x3s2_fwd(x):
    """Returns (y, idx): idx holds, per output element, the 3x3 tap (0..8) that was the maximum."""
    x = _req_bf16_nhwc(x, "x")
    n, c, hi, wi = x.shape
    if c % 8:
        raise ValueError("max-pool kernel needs C % 8 == 0")
    ho, wo = (hi - 1) // 2 + 1, (wi - 1) // 2 + 1
    y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    B.check(_lib().dp_maxpool3x3s2_fwd(B.ptr(x), B.ptr(y), B.ptr(idx), n, c, hi, wi, ho, wo, B.stream_ptr()),
            "maxpool_fwd")
    return y, idx


def maxpool3x3s2_bwd(dy, idx, in_shape):
    dy = _req_bf16_nhwc(dy, "dy")
    n, c, hi, wi = in_shape
    ho, wo = dy.shape[2:]
    if tuple(idx.shape) != (n, ho, wo, c) or idx.dtype != torch.uint8 or not idx.is_contiguous():
        raise ValueError("maxpool_bwd: idx must be the uint8 [N,Ho,Wo,C] map written by the forward")
    dx = torch.empty((n, c, hi, wi), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    B.check(_lib().dp_maxpool3x3s2_bwd(B.ptr(dy), B.ptr(idx), B.ptr(dx), n, c, hi, wi, ho, wo,
                                       B.stream_ptr()), "maxpool_bwd")
    return dx


def global_avgpool_fwd(x):
    x = _req_bf16_nhwc(x, "x")
    n, c, h, w = x.shape
    if c % 8:
        raise ValueError("global-avg-pool kernel needs C % 8 == 0")
    y = torch.empty((n, c, 1, 1), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    B.check(_lib().dp_gap_fwd(B.ptr(x), B.ptr(y), n, c, h * w, B.stream_ptr()), "gap_fwd")
    return y


def global_avgpool_bwd(dy, shape):
    n, c, h, w = shape
    dy = dy.reshape(n, c).contiguous()
    if dy.dtype != torch.bfloat16:
        raise TypeError("gap_bwd needs bf16")
    dx = torch.empty(shape, dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    B.check(_lib().dp_gap_bwd(B.ptr(dy), B.ptr(dx), n, c, h * w, B.stream_ptr()), "gap_bwd")
    return dx


# ------------------------------------------------------------------ SGD / scale
def sgd_flat(p, g, m, p_lowp, lr, momentum, weight_decay, grad_scale, zero_grad, first_step):
    for t, nm in ((p, "p"), (g, "g"), (m, "m")):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError(f"sgd_flat: {nm} must be contiguous fp32")
    if p_lowp is not None and (p_lowp.dtype != torch.bfloat16 or p_lowp.numel() != p.numel()):
        raise TypeError("sgd_flat: low-precision copy must be bf16 with the same numel")
    B.check(_lib().dp_sgd_flat(B.ptr(p), B.ptr(g), B.ptr(m), B.ptr(p_lowp), p.numel(), float(lr), float(momentum),
                               float(weight_decay), float(grad_scale), int(zero_grad), int(first_step),
                               B.stream_ptr()), "sgd_flat")


# ------------------------------------------------------------------ augmentation
def affine_augment(img_u8, lab_u8, mats, out_size, mean, std, out_dtype, pad_to: int = 3):
    if out_dtype != torch.bfloat16:
        raise TypeError("HIP augmentation writes bf16")
    img_u8 = img_u8.contiguous()
    bsz, hi, wi, _ = img_u8.shape
    ho, wo = out_size
    mats = mats.to(device=img_u8.device, dtype=torch.float32).contiguous()
    if lab_u8 is not None:
        if lab_u8.dtype != torch.uint8 or tuple(lab_u8.shape) != (bsz, hi, wi):
            raise ValueError("label must be uint8 [B,H,W]")
        lab_u8 = lab_u8.contiguous()
    cout = max(3, pad_to)
    out = torch.empty((bsz, ho, wo, cout), dtype=torch.bfloat16, device=img_u8.device)
    out_lab = torch.empty((bsz, ho, wo), dtype=torch.uint8, device=img_u8.device) if lab_u8 is not None else None
    import ctypes
    mean_c = (ctypes.c_float * 3)(*[float(v) for v in mean])
    std_c = (ctypes.c_float * 3)(*[float(v) for v in std])
    B.check(_lib().dp_affine_augment(B.ptr(img_u8), B.ptr(lab_u8), B.ptr(mats), B.ptr(out), B.ptr(out_lab),
                                     bsz, hi, wi, ho, wo, cout, mean_c, std_c, B.stream_ptr()), "affine_augment")
    return out.permute(0, 3, 1, 2), out_lab  # [B,C,H,W] view over NHWC memory (channels-last)


def _i32(t, shape, name, device):
    if t is None:
        return None
    t = t.to(device=device, dtype=torch.int32).contiguous()
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected int32 {tuple(shape)}, got {tuple(t.shape)}")
    return t


def affine_augment2(img_u8, lab_u8, src_idx, sizes, mats, clip, out_size, interp, lab_fill, clamp, mean, std,
                    border, channels, out_dtype, validated=False):
    import ctypes
    if out_dtype != torch.bfloat16:
        raise TypeError("HIP augmentation writes bf16")
    if not img_u8.is_contiguous():
        raise ValueError("affine_augment2: the sample store must be contiguous [N,H,W,3] uint8")
    nslots, hi, wi, _ = img_u8.shape
    dev = img_u8.device
    bsz = mats.shape[0]
    ho, wo = out_size
    mats = mats.to(device=dev, dtype=torch.float32).contiguous()
    src_idx = _i32(src_idx, (bsz,), "src_idx", dev)
    sizes = _i32(sizes, (bsz, 2), "sizes", dev)
    if clip is not None:
        clip = clip.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(clip.shape) != (bsz, 8):
            raise ValueError("clip must be [B,8]")
    if lab_u8 is not None and (lab_u8.dtype != torch.uint8 or tuple(lab_u8.shape) != (nslots, hi, wi)
                               or not lab_u8.is_contiguous() or lab_u8.device != dev):
        raise ValueError("label store must be contiguous uint8 [N,H,W] on the image store's device")
    # range checks of what the kernel will index with (a bad slot / size is an out-of-bounds read).  ``validated``:
    # the caller checked its HOST copies of src_idx / sizes (ops.augment.check_slots) — reading the device tensors
    # here costs a host sync per step (measured: 82 ms of "data wait" per B128 step in train_pascal.py)
    if src_idx is None and bsz != nslots:
        raise ValueError("without src_idx the store must hold exactly the batch")
    if not validated:
        if src_idx is not None:
            lo, hi_ = int(src_idx.min()), int(src_idx.max())
            if lo < 0 or hi_ >= nslots:
                raise IndexError(f"src_idx outside the store: [{lo}, {hi_}] vs {nslots} slots")
        if sizes is not None:
            if int(sizes.min()) < 1 or int(sizes[:, 0].max()) > hi or int(sizes[:, 1].max()) > wi:
                raise ValueError("sizes outside the slot dimensions")
    cout = max(3, int(channels))
    out = torch.empty((bsz, ho, wo, cout), dtype=torch.bfloat16, device=dev)
    out_lab = torch.empty((bsz, ho, wo), dtype=torch.uint8, device=dev) if lab_u8 is not None else None
    f3 = lambda v: (ctypes.c_float * 3)(*[float(x) for x in v])    # noqa: E731
    B.check(_lib().dp_affine_augment2(B.ptr(img_u8), B.ptr(lab_u8), B.ptr(src_idx), B.ptr(sizes), B.ptr(mats),
                                      B.ptr(clip), B.ptr(out), B.ptr(out_lab), bsz, hi, wi, ho, wo, cout, int(interp),
                                      int(lab_fill), int(bool(clamp)), f3(mean), f3(std), f3(border), B.stream_ptr()),
            "affine_augment2")
    return out.permute(0, 3, 1, 2), out_lab


def mask_bbox(mask_u8, src_idx, sizes, mats, validated=False):
    if mask_u8.dtype != torch.uint8 or mask_u8.dim() != 3 or not mask_u8.is_contiguous():
        raise ValueError("mask_bbox: contiguous uint8 [N,H,W]")
    nslots, hi, wi = mask_u8.shape
    dev = mask_u8.device
    bsz = mats.shape[0]
    mats = mats.to(device=dev, dtype=torch.float32).contiguous()
    src_idx = _i32(src_idx, (bsz,), "src_idx", dev)
    sizes = _i32(sizes, (bsz, 2), "sizes", dev)
    if src_idx is None and bsz != nslots:
        raise ValueError("without src_idx the store must hold exactly the batch")
    if not validated:
        if src_idx is not None and (int(src_idx.min()) < 0 or int(src_idx.max()) >= nslots):
            raise IndexError("src_idx outside the store")
        if sizes is not None and (int(sizes.min()) < 1 or int(sizes[:, 0].max()) > hi or int(sizes[:, 1].max()) > wi):
            raise ValueError("sizes outside the slot dimensions")
    bbox = torch.empty((bsz, 4), dtype=torch.int32, device=dev)
    B.check(_lib().dp_mask_bbox(B.ptr(mask_u8), B.ptr(src_idx), B.ptr(sizes), B.ptr(mats), B.ptr(bbox), bsz, hi, wi,
                                B.stream_ptr()), "mask_bbox")
    return bbox


def paste_jaccard(logits, crop_bbox, gt_u8, void_u8, src_idx, sizes, relax, thresholds, counts=None, validated=False):
    """int32 [B, T, 2] (intersection, union) — see ops.augment.paste_jaccard."""
    import ctypes
    bsz, hc, wc = logits.shape[0], logits.shape[-2], logits.shape[-1]
    if logits.numel() != bsz * hc * wc or not logits.is_contiguous() or logits.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("paste_jaccard: contiguous fp32 / bf16 logits [B,1,Hc,Wc]")
    if gt_u8.dtype != torch.uint8 or gt_u8.dim() != 3 or not gt_u8.is_contiguous():
        raise ValueError("paste_jaccard: contiguous uint8 masks [N,H,W]")
    if void_u8 is not None and (void_u8.dtype != torch.uint8 or void_u8.shape != gt_u8.shape or not void_u8.is_contiguous()):
        raise ValueError("paste_jaccard: void mask must match the object masks")
    nslots, hi, wi = gt_u8.shape
    dev = logits.device
    src_idx = _i32(src_idx, (bsz,), "src_idx", dev)
    sizes = _i32(sizes, (bsz, 2), "sizes", dev)
    bbox = _i32(crop_bbox, (bsz, 4), "crop_bbox", dev)
    if src_idx is None and bsz != nslots:
        raise ValueError("without src_idx the store must hold exactly the batch")
    if not validated:
        if src_idx is not None and (int(src_idx.min()) < 0 or int(src_idx.max()) >= nslots):
            raise IndexError("src_idx outside the store")
        if int(sizes.min()) < 1 or int(sizes[:, 0].max()) > hi or int(sizes[:, 1].max()) > wi:
            raise ValueError("sizes outside the slot dimensions")
    nt = len(thresholds)
    if not 1 <= nt <= 4:
        raise ValueError("paste_jaccard: 1..4 thresholds")
    if counts is None:
        counts = torch.zeros((bsz, nt, 2), dtype=torch.int32, device=dev)
    elif counts.dtype != torch.int32 or tuple(counts.shape) != (bsz, nt, 2) or not counts.is_contiguous():
        raise ValueError("paste_jaccard: counts must be contiguous int32 [B, T, 2]")
    th = (ctypes.c_float * nt)(*[float(t) for t in thresholds])
    B.check(_lib().dp_paste_jaccard(B.ptr(logits), int(logits.dtype == torch.bfloat16), B.ptr(gt_u8), B.ptr(void_u8),
                                    B.ptr(src_idx), B.ptr(sizes), B.ptr(bbox), B.ptr(counts), th, nt, bsz, hi, wi, hc, wc,
                                    int(relax), B.stream_ptr()), "paste_jaccard")
    return counts


def compose_crop(mats, bbox, sizes, slot_size, out_size, relax, zero_pad):
    dev = mats.device
    bsz = mats.shape[0]
    mats = mats.to(dtype=torch.float32).contiguous()
    bbox = _i32(bbox, (bsz, 4), "bbox", dev)
    sizes = _i32(sizes, (bsz, 2), "sizes", dev)
    om = torch.empty((bsz, 6), dtype=torch.float32, device=dev)
    oc = torch.empty((bsz, 8), dtype=torch.float32, device=dev)
    ob = torch.empty((bsz, 4), dtype=torch.int32, device=dev)
    B.check(_lib().dp_compose_crop(B.ptr(mats), B.ptr(bbox), B.ptr(sizes), B.ptr(om), B.ptr(oc), B.ptr(ob), bsz,
                                   int(slot_size[0]), int(slot_size[1]), int(out_size[0]), int(out_size[1]),
                                   int(relax), int(bool(zero_pad)), B.stream_ptr()), "compose_crop")
    return om, oc, ob


def extreme_points(mask_u8, bbox, random, seed):
    if mask_u8.dtype != torch.uint8 or mask_u8.dim() != 3 or not mask_u8.is_contiguous():
        raise ValueError("extreme_points: contiguous uint8 [B,H,W]")
    bsz, h, w = mask_u8.shape
    bbox = _i32(bbox, (bsz, 4), "bbox", mask_u8.device)
    pts = torch.empty((bsz, 4, 2), dtype=torch.int32, device=mask_u8.device)
    B.check(_lib().dp_extreme_points(B.ptr(mask_u8), B.ptr(bbox), B.ptr(pts), bsz, h, w, int(bool(random)),
                                     int(seed) & 0x7FFFFFFF, B.stream_ptr()), "extreme_points")
    return pts


def nellipse_guidance(out, pts, channel, alpha, sigma, mean, std):
    if out.dtype != torch.bfloat16 or out.dim() != 4 or not out.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("nellipse_guidance: channels-last bf16 [B,C,H,W]")
    bsz, c, h, w = out.shape
    pts = _i32(pts, (bsz, 4, 2), "pts", out.device)
    red = torch.empty((bsz, 2), dtype=torch.int32, device=out.device)
    B.check(_lib().dp_nellipse_guidance(B.ptr(pts), B.ptr(red), B.ptr(out), bsz, h, w, c, int(channel), float(alpha),
                                        float(sigma), float(mean), float(std), B.stream_ptr()), "nellipse_guidance")
    return out


# ------------------------------------------------------------------ confusion matrix
def confusion(pred, target, num_classes, ignore_index, out):
    if out.dtype != torch.int64 or out.numel() != num_classes * num_classes or not out.is_contiguous():
        raise ValueError("confusion: out must be contiguous int64 [K,K]")
    if target.dtype not in (torch.uint8, torch.int64):
        target = target.long()
    target = target.contiguous()
    tk = 0 if target.dtype == torch.uint8 else 2
    if pred.dim() == 4:
        pred = _req_bf16_nhwc(pred, "logits")
        n, c, h, w = pred.shape
        m, kind = n * h * w, 0
    else:
        if pred.dtype not in (torch.uint8, torch.int64):
            pred = pred.long()
        pred = pred.contiguous()
        c, m = 0, pred.numel()
        kind = 1 if pred.dtype == torch.uint8 else 2
    if target.numel() != m:
        raise ValueError("confusion: pred/target size mismatch")
    B.check(_lib().dp_confusion(B.ptr(pred), kind, B.ptr(target), tk, m, c, num_classes, int(ignore_index),
                                B.ptr(out), B.stream_ptr()), "confusion")


def pad_channels(x: torch.Tensor, cout: int) -> torch.Tensor:
    x = _req_bf16_nhwc(x, "x")
    n, c, h, w = x.shape
    y = torch.empty((n, cout, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    B.check(_lib().dp_pad_channels(B.ptr(x), B.ptr(y), n * h * w, c, cout, B.stream_ptr()), "pad_channels")
    return y


def stem_pack(x: torch.Tensor, wp: int, left: int) -> torch.Tensor:
    """channels-last bf16 [N, C<=4, H, W] -> plain [N, H, wp, 4] with ``left`` zero pixels before each
    row's data (the windowed stem convolution's input layout, csrc/misc.hip stem_pack_kernel)."""
    n, c, h, w = x.shape
    if x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("stem_pack: channels-last bf16 input")
    y = torch.empty((n, h, wp, 4), dtype=torch.bfloat16, device=x.device)
    B.check(_lib().dp_stem_pack(B.ptr(x), B.ptr(y), n * h, w, c, wp, left, B.stream_ptr()), "stem_pack")
    return y


def balanced_bce_fwd(xs, y, weights):
    """(loss fp32 scalar view, coeff fp32 [7]) — see ops.loss.balanced_bce_multi."""
    h, m = len(xs), xs[0].numel()
    lib = _lib()
    part = torch.empty(int(lib.dp_bce_blocks(m)) * 8, dtype=torch.float32, device=y.device)
    out = torch.empty(7, dtype=torch.float32, device=y.device)
    p = [B.ptr(t) for t in xs] + [None] * (3 - h)
    w = [float(v) for v in weights[:h]] + [0.0] * (3 - h)
    B.check(lib.dp_balanced_bce_fwd(p[0], p[1], p[2], h, int(xs[0].dtype == torch.bfloat16), B.ptr(y), m,
                                    int(xs[0].shape[0]), w[0], w[1], w[2], B.ptr(part), B.ptr(out), B.stream_ptr()),
            "balanced_bce_fwd")
    return out


def balanced_bce_bwd(xs, y, coeff, gloss):
    h, m = len(xs), xs[0].numel()
    grads = [torch.empty_like(t) for t in xs]
    p = [B.ptr(t) for t in xs] + [None] * (3 - h)
    d = [B.ptr(t) for t in grads] + [None] * (3 - h)
    B.check(_lib().dp_balanced_bce_bwd(p[0], p[1], p[2], h, int(xs[0].dtype == torch.bfloat16), B.ptr(y), m, B.ptr(coeff),
                                       B.ptr(gloss), d[0], d[1], d[2], B.stream_ptr()), "balanced_bce_bwd")
    return grads
