"""Ignore-index softmax cross-entropy, SURVEY.md §2.5 K7.

The reference computes its loss with ``criterion.forward(output, gts)``
(`/root/reference/train_pascal.py:199`); the north star replaces the binary loss by a 21-class
softmax CE with the VOC void label 255 ignored.  GPU path: the forward kernel reduces the per-pixel
loss and the valid-pixel count on device; the backward kernel recomputes the softmax from the
logits and writes ``dlogits`` already scaled by ``grad/count`` (3 passes over the logits in total,
nothing but the logits held between forward and backward).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _backend as B


class _SoftmaxCEHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        from . import _hip_misc
        loss, sums, logits_nhwc, target_c = _hip_misc.softmax_ce_fwd(logits, target, ignore_index)
        ctx.save_for_backward(logits_nhwc, target_c, sums)
        ctx.ignore_index = ignore_index
        return loss

    @staticmethod
    def backward(ctx, gloss):
        from . import _hip_misc
        logits, target, sums = ctx.saved_tensors
        return _hip_misc.softmax_ce_bwd(logits, target, ctx.ignore_index, sums, gloss), None, None


class _UpsampleCEHip(torch.autograd.Function):
    """CE(bilinear(logits → label size), labels) in one pass each way (csrc/upsample_ce.hip)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        from . import _hip_misc
        r = _hip_misc.upsample_ce_fwd(logits, target, ignore_index)
        if r is None:
            raise NotImplementedError("fused upsample+CE: shape not taken")
        loss, sums, x, target_c, ld = r
        ctx.save_for_backward(x, target_c, sums)
        ctx.ld = ld
        ctx.ignore_index = ignore_index
        return loss

    @staticmethod
    def backward(ctx, gloss):
        from . import _hip_misc
        x, target, sums = ctx.saved_tensors
        return _hip_misc.upsample_ce_bwd(x, ctx.ld, target, ctx.ignore_index, sums, gloss), None, None


def _fused_upsample_ok(logits, target) -> bool:
    c, hi, wi = logits.shape[1:]
    ho, wo = target.shape[1:]
    # the kernels' LDS carve (see dp_upsample_ce_fwd/_bwd): upsampling only, a bounded footprint
    if ho < hi or wo < wi or c > 64:
        return False
    fh = int(5 * (ho - 1) / max(hi - 1, 1)) + 2 if hi > 1 else ho
    fw = int(9 * (wo - 1) / max(wi - 1, 1)) + 2 if wi > 1 else wo
    return (60 * c * 4 + (fh + fw) * 8 + (4 * fh + 8 * fw) * 4 + fh * fw * c * 2 <= 64 * 1024
            and 9 * 33 * c * 4 <= 60 * 1024)


def softmax_cross_entropy(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = 255) -> torch.Tensor:
    """Mean CE over non-ignored pixels.  ``logits`` [N,C,H,W], ``target`` [N,H',W'] integer.  When
    the label map is larger than the logits, the loss is that of the logits bilinearly upsampled
    (align_corners=True) to the label size — the DeepLab head — computed without materialising them."""
    if tuple(logits.shape[2:]) != tuple(target.shape[1:]):
        if B.use_hip(logits) and _fused_upsample_ok(logits, target):
            return _UpsampleCEHip.apply(logits, target, ignore_index)
        from .upsample import bilinear_resize
        logits = bilinear_resize(logits, tuple(target.shape[1:]))
    if B.use_hip(logits):
        return _SoftmaxCEHip.apply(logits, target, ignore_index)
    # same convention as the kernel: Σ loss / max(valid pixels, 1) — an all-ignored batch gives 0, not NaN
    t = target.long()
    total = F.cross_entropy(logits.float(), t, ignore_index=ignore_index, reduction="sum")
    return total / (t != ignore_index).sum().clamp_min(1).to(total.dtype)


class SoftmaxCrossEntropy(nn.Module):
    # the engine lets models that can skip their final upsample do so (``defer_upsample``)
    fuses_upsample = True

    def __init__(self, ignore_index: int = 255):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        if isinstance(logits, (tuple, list)):  # auxiliary heads: main + 0.4·aux (FCN convention)
            main = softmax_cross_entropy(logits[0], target, self.ignore_index)
            for aux in logits[1:]:
                main = main + 0.4 * softmax_cross_entropy(aux, target, self.ignore_index)
            return main
        return softmax_cross_entropy(logits, target, self.ignore_index)


class _BalancedBceHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, weights, *xs):
        from . import _hip_misc
        xs = tuple(t.contiguous() for t in xs)
        coeff = _hip_misc.balanced_bce_fwd(xs, y, weights)
        ctx.save_for_backward(y, coeff, *xs)
        return coeff[0].clone()

    @staticmethod
    def backward(ctx, g):
        from . import _hip_misc
        y, coeff, *xs = ctx.saved_tensors
        grads = _hip_misc.balanced_bce_bwd(xs, y, coeff, g.detach().float().contiguous())
        return (None, None, *grads)


def balanced_bce_multi(outputs, target: torch.Tensor, weights) -> Optional[torch.Tensor]:
    """``sum_h w_h * class_balanced_cross_entropy_loss(outputs[h], target, size_average=False, batch_average=True)``
    for up to 3 heads in three kernel launches (csrc/loss_upsample.hip) — or ``None`` when the fused kernels do not
    take the call (CPU tensors, the stock-op backend, more than 3 heads, mixed shapes / dtypes): the caller then runs
    the tensor-op formulation (utils.losses)."""
    outputs = tuple(outputs)
    if not (1 <= len(outputs) <= 3) or not B.use_hip(outputs[0]):
        return None
    o0 = outputs[0]
    if o0.dtype not in (torch.float32, torch.bfloat16) or target.numel() != o0.numel():
        return None
    if any(o.shape != o0.shape or o.dtype != o0.dtype or o.device != o0.device for o in outputs):
        return None
    y = target.detach().reshape(o0.shape).float().contiguous()
    return _BalancedBceHip.apply(y, tuple(float(w) for w in weights), *outputs)
