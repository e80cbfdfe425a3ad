"""Aligned Xception-65 backbone (DeepLabv3+), ``BASELINE.json`` config 5 (output-stride 8,
"heavy atrous conv").

Every separable unit is ``depthwise 3×3 (atrous) → BN → pointwise 1×1 → BN``: the depthwise part
runs the memory-bound K13 kernel, the pointwise part the MFMA implicit-GEMM kernel (SURVEY.md
§2.5).  Activations between units are fused into the preceding BN (``relu=True``) wherever the
pre-activation structure allows it.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from ..ops import conv_bn_act, BatchNorm2d, Conv2d
from ..ops.sepconv import sepconv_bn_act


class SeparableConv2d(nn.Module):
    """'same'-padded depthwise (dilated) conv + BN + pointwise conv."""

    def __init__(self, cin: int, cout: int, stride: int = 1, dilation: int = 1):
        super().__init__()
        # effective kernel 3 + 2·(d-1); symmetric pad = d keeps 'same' for odd maps (513-derived)
        self.conv1 = Conv2d(cin, cin, 3, stride=stride, padding=dilation, dilation=dilation,
                            groups=cin, bias=False)
        self.bn = BatchNorm2d(cin)
        self.pointwise = Conv2d(cin, cout, 1, bias=False)

    def depthwise_bn(self, x):
        return conv_bn_act(self.conv1, self.bn, x)

    def forward(self, x):
        return self.pointwise(self.depthwise_bn(x))


class _Unit(nn.Module):
    """separable conv + BN (+ fused ReLU feeding the next unit)."""

    def __init__(self, cin, cout, stride, dilation, relu_after):
        super().__init__()
        self.sep = SeparableConv2d(cin, cout, stride, dilation)
        self.bn = BatchNorm2d(cout)
        self.relu_after = relu_after

    def forward(self, x, residual=None, relu=None, passthrough=False, sole_consumer=False):
        # (GPU training path: the depthwise BatchNorm is folded into the pointwise weights, ops.sepconv)
        return sepconv_bn_act(self.sep, self.bn, x, relu=self.relu_after if relu is None else relu,
                              residual=residual, passthrough=passthrough, sole_consumer=sole_consumer)


class XceptionBlock(nn.Module):
    def __init__(self, cin: int, cout: int, reps: int, stride: int = 1, dilation: int = 1,
                 start_with_relu: bool = True, grow_first: bool = True, is_last: bool = False):
        super().__init__()
        self.start_with_relu = start_with_relu
        # set by AlignedXception: every consumer of this block's output applies a ReLU first, so the block
        # emits relu(out) from its last fused BN kernel (BN + shortcut + ReLU in one pass: no add kernel,
        # no separate ReLU forward / backward over the 728-channel maps) ...
        self.activate_output = False
        # ... and the block that receives such an input does not activate it again
        self.input_activated = False
        # set by AlignedXception: nothing but this block consumes its input tensor
        self.input_sole = False
        if cout != cin or stride != 1:
            self.skip = Conv2d(cin, cout, 1, stride=stride, bias=False)
            self.skipbn = BatchNorm2d(cout)
        else:
            self.skip = None
        plan = []  # (cin, cout, stride, dilation)
        c = cin
        if grow_first:
            plan.append((c, cout, 1, dilation))
            c = cout
        for _ in range(reps - 1):
            plan.append((c, c, 1, dilation))
        if not grow_first:
            plan.append((c, cout, 1, dilation))
        if stride != 1:
            plan.append((cout, cout, stride, 1))
        elif is_last:
            plan.append((cout, cout, 1, dilation))
        units = []
        for i, (a, b, s, d) in enumerate(plan):
            units.append(_Unit(a, b, s, d, relu_after=(i + 1 < len(plan))))
        self.rep = nn.Sequential(*units)

    def forward(self, inp):
        x = torch.relu(inp) if (self.start_with_relu and not self.input_activated) else inp
        # `x` is a fused BN+ReLU output consumed by nothing but this block: its BatchNorm's backward reductions
        # ride in the first backward kernel that sees the block's whole input gradient (ops.sepconv / ops.conv)
        sole = self.input_sole and x is inp
        units = list(self.rep)
        # pre-activation quirk of the published weights: the shortcut sees the activated input
        if self.skip is not None:
            if self.skipbn.training and not self.skipbn.frozen and self.skip.can_fuse_stats(x) \
                    and self.skip.can_passthrough(x):
                # the shortcut convolution first, the units on its passthrough: the units' input gradient is added in
                # the shortcut's data-gradient store loop (no add kernel)
                sk, st, x = self.skip(x, with_stats=True, passthrough=True, sole_consumer=sole)
                for i, unit in enumerate(units):
                    x = unit(x, sole_consumer=i > 0)
                return self.skipbn(sk, relu=self.activate_output, residual=x, stats=st)
            short = x
            for i, unit in enumerate(units):
                x = unit(x, sole_consumer=i > 0)
            return conv_bn_act(self.skip, self.skipbn, short, residual=x, relu=self.activate_output)
        x, short = units[0](x, passthrough=True, sole_consumer=sole)
        for unit in units[1:-1]:
            x = unit(x, sole_consumer=True)
        return units[-1](x, residual=short, relu=self.activate_output, sole_consumer=True)


class AlignedXception(nn.Module):
    """Returns ``(features[2048], low_level[128 @ stride 4])``."""

    def __init__(self, output_stride: int = 16, in_channels: int = 3):
        super().__init__()
        if output_stride == 16:
            entry3_stride, mid_dil, exit_dil = 2, 1, (1, 2)
        elif output_stride == 8:
            entry3_stride, mid_dil, exit_dil = 1, 2, (2, 4)
        else:
            raise ValueError("output_stride must be 8 or 16")
        self.conv1 = Conv2d(in_channels, 32, 3, stride=2, padding=1, bias=False)
        self.bn1 = BatchNorm2d(32)
        self.conv2 = Conv2d(32, 64, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(64)
        self.block1 = XceptionBlock(64, 128, 2, stride=2, start_with_relu=False)
        self.block2 = XceptionBlock(128, 256, 2, stride=2, start_with_relu=False)
        self.block3 = XceptionBlock(256, 728, 2, stride=entry3_stride, is_last=True)
        self.middle = nn.Sequential(*[XceptionBlock(728, 728, 3, dilation=mid_dil) for _ in range(16)])
        self.block20 = XceptionBlock(728, 1024, 2, dilation=exit_dil[0], grow_first=False, is_last=True)
        self.conv3 = SeparableConv2d(1024, 1536, dilation=exit_dil[1])
        self.bn3 = BatchNorm2d(1536)
        self.conv4 = SeparableConv2d(1536, 1536, dilation=exit_dil[1])
        self.bn4 = BatchNorm2d(1536)
        self.conv5 = SeparableConv2d(1536, 2048, dilation=exit_dil[1])
        self.bn5 = BatchNorm2d(2048)
        self.out_channels, self.low_level_channels = 2048, 128
        self.output_stride = output_stride
        # every block output is consumed through a ReLU (the next block's pre-activation, `low`, the
        # ReLU before conv3): produce it activated
        chain = [self.block1, self.block2, self.block3, *self.middle, self.block20]
        for blk in chain:
            blk.activate_output = True
        for blk in chain[1:]:
            blk.input_activated = True
        # block1's input (conv2 + BN + ReLU) and every block output except block1's (`low` also feeds the decoder)
        # have exactly one consumer
        for blk in chain:
            blk.input_sole = blk is not self.block2

    def forward(self, x):
        x = conv_bn_act(self.conv1, self.bn1, x, relu=True)
        x = conv_bn_act(self.conv2, self.bn2, x, relu=True)
        low = self.block1(x)              # = relu(block output): see activate_output
        x = self.block2(low)
        x = self.block3(x)
        x = self.middle(x)
        x = self.block20(x)
        x = sepconv_bn_act(self.conv3, self.bn3, x, relu=True, sole_consumer=True)
        x = sepconv_bn_act(self.conv4, self.bn4, x, relu=True, sole_consumer=True)
        x = sepconv_bn_act(self.conv5, self.bn5, x, relu=True, sole_consumer=True)
        return x, low
