"""Pooling kernels, SURVEY.md §2.5 K11: 3×3/s2 max-pool (ResNet stem) and global average pool
(ASPP image branch).  NHWC bf16 on GPU; the backward of max-pool recomputes the arg-max from the
saved input instead of storing an index tensor."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _backend as B


class _MaxPool3x3s2Hip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from . import _hip_misc
        y, idx = _hip_misc.maxpool3x3s2_fwd(x)
        ctx.save_for_backward(idx)
        ctx.in_shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _hip_misc
        (idx,) = ctx.saved_tensors
        return _hip_misc.maxpool3x3s2_bwd(dy, idx, ctx.in_shape)


class _GlobalAvgPoolHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from . import _hip_misc
        ctx.shape = x.shape
        return _hip_misc.global_avgpool_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        from . import _hip_misc
        return _hip_misc.global_avgpool_bwd(dy, ctx.shape)


def max_pool3x3s2(x: torch.Tensor) -> torch.Tensor:
    """kernel 3, stride 2, padding 1."""
    if B.use_hip(x):
        return _MaxPool3x3s2Hip.apply(x)
    return F.max_pool2d(x, 3, 2, 1)


def global_avg_pool(x: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W] -> [N,C,1,1]."""
    if B.use_hip(x):
        return _GlobalAvgPoolHip.apply(x)
    return F.adaptive_avg_pool2d(x, 1)


class MaxPool3x3s2(nn.Module):
    def forward(self, x):
        return max_pool3x3s2(x)


class GlobalAvgPool(nn.Module):
    def forward(self, x):
        return global_avg_pool(x)
