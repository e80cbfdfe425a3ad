"""Convolution: implicit-GEMM (dilated) conv on MFMA for gfx950, plus depthwise conv.

SURVEY.md §2.5 K1/K2/K3 (dense conv fwd / dgrad / wgrad) and K13 (depthwise atrous conv for the
Xception backbone).  The reference only ever reaches a convolution through its external model
(`/root/reference/train_pascal.py:86`, `:195`, `:216`); the kernels here are what that call lowers
to on MI355X.

Layout contract of the GPU path: activations are ``[N, C, H, W]``-shaped tensors stored
channels-last (NHWC in memory), bf16; weights are ``[Cout, Cin/g, kh, kw]`` fp32 master parameters
(state_dict compatible) with a cached bf16 channels-last copy (KRSC in memory) consumed by the
kernels.  Gradients w.r.t. weights are produced in fp32.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _backend as B

_IntOr2 = Union[int, Tuple[int, int]]


def _pair(v: _IntOr2) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def conv_out_size(n: int, k: int, s: int, p: int, d: int) -> int:
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


# --------------------------------------------------------------------------------------
# functional entry
# --------------------------------------------------------------------------------------
def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           stride: _IntOr2 = 1, padding: _IntOr2 = 0, dilation: _IntOr2 = 1, groups: int = 1,
           w_lowp: Optional[torch.Tensor] = None, with_stats: bool = False, passthrough: bool = False,
           sole_consumer: bool = False):
    """2-D convolution.  ``w_lowp`` is an optional up-to-date bf16 KRSC copy of ``weight``.

    ``with_stats`` (GPU, dense conv only) additionally returns the per-channel ``[R, 2, Cout]``
    partial sums Σy, Σy² accumulated in the kernel epilogue — the BatchNorm statistics for free."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    if B.use_hip(x):
        from . import _hip_conv
        return _hip_conv.conv2d(x, weight, bias, stride, padding, dilation, groups, w_lowp, with_stats,
                                passthrough, sole_consumer)
    if with_stats:
        raise ValueError("with_stats is a GPU-kernel feature")
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = None if bias is None else bias.to(x.dtype)
    y = F.conv2d(x, w, b, stride, padding, dilation, groups)
    return (y, x) if passthrough else y


class Conv2d(nn.Module):
    """Drop-in ``nn.Conv2d`` (same parameter names/shapes) running the HIP kernels on GPU."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: _IntOr2, stride: _IntOr2 = 1,
                 padding: _IntOr2 = 0, dilation: _IntOr2 = 1, groups: int = 1, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.groups = groups
        kh, kw = self.kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kh, kw))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._lowp: Optional[torch.Tensor] = None
        self._lowp_version = -1
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        if self.bias is not None:
            fan_in = self.weight[0].numel()
            bound = 1.0 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def lowp_weight(self, dtype: torch.dtype) -> torch.Tensor:
        """bf16 KRSC copy of the master weight, refreshed when the parameter changes.

        The fused SGD kernel (ops.sgd) writes this copy in the same pass as the fp32 update and
        bumps ``_lowp_version`` so no separate cast kernel runs in steady state."""
        w = self.weight
        arena_copy = getattr(w, "_dpamd_lowp", None)
        if arena_copy is not None and arena_copy.dtype == dtype:
            # parallel.arena keeps this view fresh (the fused SGD kernel writes it); only an
            # out-of-band in-place edit of the master weight (init, load_state_dict) needs a re-cast
            if getattr(w, "_dpamd_lowp_version", -1) != w._version:
                with torch.no_grad():
                    arena_copy.copy_(w.detach())
                w._dpamd_lowp_version = w._version
                w._dpamd_lowpT_ok = False      # backward re-transposes this layer itself
            return arena_copy
        if (self._lowp is None or self._lowp_version != w._version or self._lowp.dtype != dtype
                or self._lowp.device != w.device):
            with torch.no_grad():
                if (self._lowp is not None and self._lowp.dtype == dtype and self._lowp.device == w.device
                        and self._lowp.shape == w.shape):
                    self._lowp.copy_(w.detach())      # same storage: captured graphs keep a valid pointer
                else:
                    self._lowp = w.detach().to(dtype).contiguous(memory_format=torch.channels_last)
            self._lowp_version = w._version
        return self._lowp

    def forward(self, x: torch.Tensor, with_stats: bool = False, passthrough: bool = False,
                sole_consumer: bool = False):
        """``sole_consumer``: the caller guarantees that ``x`` (a fused BN+ReLU output) feeds nothing
        but this convolution (and its ``passthrough``), so the whole gradient of ``x`` leaves this
        node and BN's backward prologue can run inside the data-gradient kernel."""
        w_lowp = self.lowp_weight(x.dtype) if (B.use_hip(x) and x.dtype != self.weight.dtype) else None
        return conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation,
                      self.groups, w_lowp, with_stats, passthrough, sole_consumer)

    def can_passthrough(self, x: torch.Tensor) -> bool:
        depthwise = (self.groups == self.in_channels == self.out_channels and self.kernel_size == (3, 3)
                     and self.bias is None)
        return (B.use_hip(x) and (self.groups == 1 or depthwise) and self.in_channels % 8 == 0
                and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and x.requires_grad)

    def can_fuse_stats(self, x: torch.Tensor) -> bool:
        dense = self.groups == 1
        depthwise = (self.groups == self.in_channels == self.out_channels and self.kernel_size == (3, 3)
                     and self.bias is None)
        return (B.use_hip(x) and (dense or depthwise) and self.out_channels % 8 == 0
                and x.dtype == torch.bfloat16)

    def extra_repr(self) -> str:
        s = f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}"
        if self.padding != (0, 0):
            s += f", padding={self.padding}"
        if self.dilation != (1, 1):
            s += f", dilation={self.dilation}"
        if self.groups != 1:
            s += f", groups={self.groups}"
        if self.bias is None:
            s += ", bias=False"
        return s


INFER_FUSE = True         # (module switch for tests / tools: False = convolution and evaluation BatchNorm as two passes)


def _infer_fused(conv: "Conv2d", bn, x: torch.Tensor, relu: bool, residual: Optional[torch.Tensor]):
    from . import _hip_conv
    n, _, hi, wi = x.shape
    r, s = conv.kernel_size
    ho = _hip_conv._out_size(hi, r, conv.stride[0], conv.padding[0], conv.dilation[0])
    wo = _hip_conv._out_size(wi, s, conv.stride[1], conv.padding[1], conv.dilation[1])
    res = _hip_conv.as_nhwc(residual) if residual is not None else None
    return _hip_conv.igemm_infer(_hip_conv.as_nhwc(x), conv.lowp_weight(x.dtype), (ho, wo), conv.stride[0], 1,
                                 tuple(conv.padding), tuple(conv.dilation), bn.eval_affine(), relu, res)


def conv_bn_act(conv: "Conv2d", bn, x: torch.Tensor, relu: bool = False,
                residual: Optional[torch.Tensor] = None, passthrough: bool = False, sole_consumer: bool = False,
                dropout: float = 0.0, out=None):
    """``act(BN(conv(x)) + residual)``.  On the GPU path the batch statistics come out of the conv
    kernel's epilogue (no separate statistics pass over the activation).

    ``dropout``: probability of a dropout behind the activation (training only), run inside the BN
    kernels on the GPU path (ops.batchnorm).

    ``out``: an ``ops.concat`` slot — the result is written straight into that channel slice of a concat
    buffer (GPU path; ignored elsewhere, ``ChannelConcat.result`` then falls back to ``torch.cat``).

    ``passthrough=True`` returns ``(out, x')`` where ``x'`` is ``x`` routed through the convolution's
    autograd node: use ``x'`` as the block's identity and the skip-path gradient is added inside the
    data-gradient kernel (csrc/conv_igemm2.hip store loop) instead of by an add kernel."""
    slot = out
    if (INFER_FUSE and not bn.training and slot is None and not torch.is_grad_enabled() and B.use_hip(x)
            and x.dtype == torch.bfloat16 and conv.groups == 1 and conv.bias is None
            and conv.stride[0] == conv.stride[1] and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0):
        # evaluation: BatchNorm (running statistics), identity add and ReLU in the convolution's epilogue
        y = _infer_fused(conv, bn, x, relu, residual)
        if y is not None:
            return (y, x) if passthrough else y
    pt = passthrough and conv.can_passthrough(x)
    if bn.training and not bn.frozen and conv.can_fuse_stats(x):
        if pt:
            y, stats, skip = conv(x, with_stats=True, passthrough=True, sole_consumer=sole_consumer)
        else:
            y, stats = conv(x, with_stats=True, sole_consumer=sole_consumer)
        out = bn(y, relu=relu, residual=residual, stats=stats, dropout=dropout, out=slot)
    elif pt:
        y, skip = conv(x, passthrough=True, sole_consumer=sole_consumer)
        out = bn(y, relu=relu, residual=residual, dropout=dropout, out=slot)
    else:
        out = bn(conv(x, sole_consumer=sole_consumer), relu=relu, residual=residual, dropout=dropout, out=slot)
    if passthrough:
        return out, (skip if pt else x)
    return out
