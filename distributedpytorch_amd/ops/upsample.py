"""Bilinear up/down-sampling (``align_corners=True``), SURVEY.md §2.5 K6.

The reference's (commented) call is ``upsample(output, size=(512, 512), mode='bilinear',
align_corners=True)`` (`/root/reference/train_pascal.py:196`, `:246`); DeepLabv3+ uses the same op
for the ASPP→decoder (33²→129²) and logits→input (129²→513²) resizes.  GPU path: gather forward
and *scatter-free* backward (every low-res pixel gathers its contributors) on NHWC bf16.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _backend as B


class _BilinearHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, out=None):
        from . import _hip_misc
        ctx.in_size = (x.shape[2], x.shape[3])
        return _hip_misc.bilinear_fwd(x, size, out)

    @staticmethod
    def backward(ctx, dy):
        from . import _hip_misc
        return _hip_misc.bilinear_bwd(dy, ctx.in_size), None, None


def bilinear_resize(x: torch.Tensor, size: Tuple[int, int], out=None) -> torch.Tensor:
    """``out``: an ``ops.concat`` slot to write the result into (GPU path; ignored elsewhere)."""
    size = (int(size[0]), int(size[1]))
    if tuple(x.shape[2:]) == size:
        return x
    if B.use_hip(x):
        return _BilinearHip.apply(x, size, out)
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


class Upsample(nn.Module):
    def __init__(self, size: Tuple[int, int]):
        super().__init__()
        self.size = size

    def forward(self, x):
        return bilinear_resize(x, self.size)
