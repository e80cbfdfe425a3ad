"""Dilated ResNet backbones (18/34/50/101) for segmentation.

The reference builds its network from an external package (``DANet(1, 'resnet101')``,
`/root/reference/train_pascal.py:86`) or the commented DeepLab ctor
``deeplab.DeepLab(backbone='resnet', output_stride=16, ...)`` (`:85`).  This is a fresh
implementation on the fused ops layer: every ``conv → BN → ReLU`` is one implicit-GEMM launch plus
one fused BN/activation launch, and the bottleneck's ``BN → +identity → ReLU`` tail is a single
kernel (``BatchNorm2d.forward(x, relu=True, residual=identity)``).

Parameter names follow the usual ``conv1/bn1/layer{1..4}.{i}.conv{j}/bn{j}/downsample.{0,1}``
scheme so published ResNet / DeepLab ``state_dict`` files map 1:1.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from ..ops import conv_bn_act, BatchNorm2d, Conv2d, MaxPool3x3s2
from ..ops.bnfold import conv_bn_add_relu


class _Projection(nn.Module):
    """1×1 strided conv + BN shortcut with ``downsample.0`` / ``downsample.1`` key names."""

    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.add_module("0", Conv2d(cin, cout, 1, stride=stride, bias=False))
        self.add_module("1", BatchNorm2d(cout))

    def forward(self, x, passthrough: bool = False):
        return conv_bn_act(getattr(self, "0"), getattr(self, "1"), x, passthrough=passthrough)


class _FanOutMixin:
    """First block of a stage: ``x`` feeds conv1 AND the projection shortcut (and, for the stride-4
    map, the decoder).  Autograd would sum those gradients with one add kernel per extra consumer —
    a full read-read-write of the largest activations.  Instead ``x`` is threaded through the
    consumers' autograd nodes (``passthrough``): each data-gradient kernel adds the gradient that
    arrived from the consumers after it in its store loop, so the sum costs no launch at all."""

    # set by ResNet for the block whose input another module consumes too: forward leaves the
    # threaded alias of its input in ``input_alias`` for that consumer to pick up
    export_input = False
    input_alias: Optional[torch.Tensor] = None

    def _projected(self, x):
        # x reaches its other consumers only through conv1's autograd node, so conv1's data-gradient
        # kernel sees the WHOLE gradient of x: the producer's BN-backward prologue can run there too
        out, xp = conv_bn_act(self.conv1, self.bn1, x, relu=True, passthrough=True, sole_consumer=True)
        if self.export_input:
            identity, xp = self.downsample(xp, passthrough=True)
            self.input_alias = xp
        else:
            identity = self.downsample(xp)
        return out, identity

    def take_input_alias(self, default: torch.Tensor) -> torch.Tensor:
        alias, self.input_alias = self.input_alias, None
        return default if alias is None else alias


class BasicBlock(_FanOutMixin, nn.Module):
    expansion = 1
    # set by ResNet._stage for blocks 1..n-1 of a stage: the block's input is the previous block's
    # output and nothing else reads it
    input_exclusive = False

    def __init__(self, cin: int, planes: int, stride: int = 1, dilation: int = 1,
                 downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = Conv2d(cin, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        if self.downsample is None:
            # identity routed through conv1's autograd node: its gradient joins conv1's dgrad kernel
            # (and conv1 is then the only consumer of x: BN's backward prologue runs in that kernel too)
            out, identity = conv_bn_act(self.conv1, self.bn1, x, relu=True, passthrough=True,
                                        sole_consumer=self.input_exclusive)
        else:
            out, identity = self._projected(x)
        return conv_bn_act(self.conv2, self.bn2, out, relu=True, residual=identity, sole_consumer=True)


class Bottleneck(_FanOutMixin, nn.Module):
    expansion = 4
    input_exclusive = False      # see BasicBlock
    # set by ResNet: the block's output feeds a convolution that runs this block's ReLU mask and Σdm in its
    # data-gradient store loop (the next block's conv1) -> the last BatchNorm's backward is folded into conv3's
    # gradients (ops.bnfold) and the 4x-wide pre-BN tensor is neither kept nor re-read in backward
    fold_tail = False

    def __init__(self, cin: int, planes: int, stride: int = 1, dilation: int = 1,
                 downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = Conv2d(cin, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        if self.downsample is None:
            # identity routed through conv1's autograd node: its gradient joins conv1's dgrad kernel
            # (and conv1 is then the only consumer of x: BN's backward prologue runs in that kernel too)
            out, identity = conv_bn_act(self.conv1, self.bn1, x, relu=True, passthrough=True,
                                        sole_consumer=self.input_exclusive)
        else:
            out, identity = self._projected(x)
        out = conv_bn_act(self.conv2, self.bn2, out, relu=True, sole_consumer=True)
        if self.fold_tail:
            return conv_bn_add_relu(self.conv3, self.bn3, out, identity, sole_consumer=True)
        return conv_bn_act(self.conv3, self.bn3, out, relu=True, residual=identity, sole_consumer=True)


_CFG = {
    "resnet18": (BasicBlock, (2, 2, 2, 2)),
    "resnet34": (BasicBlock, (3, 4, 6, 3)),
    "resnet50": (Bottleneck, (3, 4, 6, 3)),
    "resnet101": (Bottleneck, (3, 4, 23, 3)),
}


def _stage_plan(output_stride: int) -> Tuple[Sequence[int], Sequence[int]]:
    if output_stride == 32:
        return (1, 2, 2, 2), (1, 1, 1, 1)
    if output_stride == 16:
        return (1, 2, 2, 1), (1, 1, 1, 2)
    if output_stride == 8:
        return (1, 2, 1, 1), (1, 1, 2, 4)
    raise ValueError("output_stride must be 8, 16 or 32")


class ResNet(nn.Module):
    """Returns ``(c5, c2)``: the stride-``output_stride`` feature map and the stride-4 low-level map.

    ``multi_grid`` applies per-block dilation multipliers in the last stage (DeepLab's (1, 2, 4))."""

    def __init__(self, arch: str = "resnet101", output_stride: int = 16, in_channels: int = 3,
                 multi_grid: Optional[Sequence[int]] = (1, 2, 4)):
        super().__init__()
        block, depths = _CFG[arch]
        strides, dilations = _stage_plan(output_stride)
        self.arch, self.output_stride = arch, output_stride
        self.inplanes = 64
        self.conv1 = Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.maxpool = MaxPool3x3s2()
        self.layer1 = self._stage(block, 64, depths[0], strides[0], dilations[0])
        self.layer2 = self._stage(block, 128, depths[1], strides[1], dilations[1])
        self.layer2[0].export_input = self.layer2[0].downsample is not None
        self.layer3 = self._stage(block, 256, depths[2], strides[2], dilations[2])
        grid = None
        if multi_grid is not None and block is Bottleneck and len(multi_grid) == depths[3]:
            grid = list(multi_grid)
        self.layer4 = self._stage(block, 512, depths[3], strides[3], dilations[3], grid)
        self.out_channels = 512 * block.expansion
        self.low_level_channels = 64 * block.expansion
        if block is Bottleneck:
            chain = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]
            for b in chain[:-1]:             # (the last block's output goes to a head with several consumers)
                b.fold_tail = True
        for m in self.modules():
            if isinstance(m, BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _stage(self, block, planes: int, depth: int, stride: int, dilation: int,
               grid: Optional[List[int]] = None) -> nn.Sequential:
        proj = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            proj = _Projection(self.inplanes, planes * block.expansion, stride)
        g = grid or [1] * depth
        blocks = [block(self.inplanes, planes, stride, dilation * g[0], proj)]
        self.inplanes = planes * block.expansion
        for i in range(1, depth):
            blocks.append(block(self.inplanes, planes, 1, dilation * g[i]))
            blocks[-1].input_exclusive = True
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = conv_bn_act(self.conv1, self.bn1, x, relu=True)
        x = self.maxpool(x)
        c2 = self.layer1(x)
        x = self.layer2(c2)
        # the decoder reads the stride-4 map too: hand it the alias threaded through layer2's first block
        c2 = self.layer2[0].take_input_alias(c2)
        x = self.layer3(x)
        c5 = self.layer4(x)
        return c5, c2
