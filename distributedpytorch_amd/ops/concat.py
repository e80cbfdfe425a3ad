"""Channel concatenation without a cat kernel.

``torch.cat`` over NHWC activations is a full read + write of every part, and its backward hands the
consumers non-contiguous slices that they first copy (DeepLabv3+ at batch 128: 1.7 ms of cat kernels
and 1.4 ms of slice copies per step on the ASPP and decoder concats,
``profiles/r1_native_v12_b128.md``).  Here the concat buffer is allocated first and every PRODUCER
(fused BN+ReLU, bilinear resize, the broadcast of the image-pooling branch) writes its channel slice in
place through an output row stride; in backward the consumers read their slice of the concat gradient
through the same stride.  ``result()`` ties the parts together for autograd at zero cost.

Producers that cannot write in place (CPU tensors, the stock-op backend, an unfused dropout) simply
return an ordinary tensor; ``result()`` notices and falls back to ``torch.cat``.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

import os

from . import _backend as B

ENABLED = True

Slot = Tuple[torch.Tensor, int, int]          # (buffer, channel offset, channels)


def row_stride(t: torch.Tensor) -> Optional[int]:
    """Channel stride ``ld`` of a [N,C,H,W] tensor whose memory is NHWC rows of ``ld >= C`` elements
    (a channels-last tensor or a channel slice of one), or None."""
    if t.dim() != 4:
        return None
    n, c, h, w = t.shape
    ld = t.stride(3)
    if w == 1:
        ld = t.stride(2)
    if h == 1 and w == 1:
        ld = t.stride(0)
    ok = (t.stride(1) == 1 or c == 1) and ld >= c and (w == 1 or t.stride(3) == ld) \
        and (h == 1 or t.stride(2) == w * ld) and (n == 1 or t.stride(0) == h * w * ld)
    return ld if ok else None


class _Assemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf, offsets, *parts):
        ctx.offsets = offsets
        ctx.widths = [p.shape[1] for p in parts]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(g[:, o:o + c] for o, c in zip(ctx.offsets, ctx.widths))


class _BroadcastInto(torch.autograd.Function):
    """[N, C, 1, 1] -> its slot of the buffer, every pixel (align_corners resize of a 1x1 map)."""

    @staticmethod
    def forward(ctx, v, buf, off):
        c = v.shape[1]
        y = buf[:, off:off + c]
        # a raw write like the kernels' (they go through pointers): the buffer's version counter must not
        # move, or autograd takes the sibling slices — outputs of other custom Functions — for stale views
        with torch.autograd._unsafe_preserve_version_counter(buf):
            y.copy_(v)
        return y

    @staticmethod
    def backward(ctx, g):
        return g.sum(dim=(2, 3), keepdim=True, dtype=torch.float32).to(g.dtype), None, None


class ChannelConcat:
    def __init__(self, like: torch.Tensor, channels: Sequence[int], hw: Tuple[int, int]):
        self.channels = [int(c) for c in channels]
        self.offsets = [sum(self.channels[:i]) for i in range(len(self.channels))]
        self.hw = (int(hw[0]), int(hw[1]))
        self.n = like.shape[0]
        total = sum(self.channels)
        # in place only where the kernels can: bf16 HIP path, 16-byte aligned slices
        self.in_place = (ENABLED and B.use_hip(like) and like.dtype == torch.bfloat16 and total % 8 == 0
                         and all(o % 8 == 0 for o in self.offsets))
        self.buf = None
        if self.in_place:
            self.buf = torch.empty((self.n, total, *self.hw), dtype=like.dtype, device=like.device,
                                   memory_format=torch.channels_last)

    def slot(self, i: int) -> Optional[Slot]:
        """Pass as ``out=`` to a producer (``conv_bn_act``, ``bilinear_resize``); None = not in place."""
        return (self.buf, self.offsets[i], self.channels[i]) if self.in_place else None

    def broadcast(self, i: int, v: torch.Tensor) -> torch.Tensor:
        if not self.in_place:
            return v.expand(-1, -1, *self.hw)
        return _BroadcastInto.apply(v, self.buf, self.offsets[i])

    def result(self, *parts: torch.Tensor) -> torch.Tensor:
        assert len(parts) == len(self.channels)
        if self.in_place:
            esz = self.buf.element_size()
            aliased = all(p.shape[1] == c and p.data_ptr() == self.buf.data_ptr() + o * esz
                          and tuple(p.shape[2:]) == self.hw and row_stride(p) == self.buf.shape[1]
                          for p, o, c in zip(parts, self.offsets, self.channels))
            if aliased:
                return _Assemble.apply(self.buf, self.offsets, *parts)
        return torch.cat(parts, dim=1)
