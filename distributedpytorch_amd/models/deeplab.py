"""DeepLabv3+ (ASPP + decoder), the headline model of ``BASELINE.json`` (R101 @ 513², OS 16).

The reference names it only in a commented constructor,
``deeplab.DeepLab(backbone='resnet', output_stride=16, num_classes=1, sync_bn=False,
freeze_bn=True, gate=4)`` (`/root/reference/train_pascal.py:85`), plus the 1×/10× learning-rate
parameter-group hint (`:90-91`).  ``get_1x_lr_params`` / ``get_10x_lr_params`` are provided with
those names.  Child-module names (``backbone``, ``aspp.aspp{1..4}.atrous_conv/bn``,
``aspp.global_avg_pool.{1,2}``, ``decoder.conv1/bn1/last_conv.{0,1,4,5,8}``) follow the widely
published DeepLab-Xception/ResNet checkpoints so those ``state_dict`` files load.
"""
from __future__ import annotations

from typing import Iterator, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import conv_bn_act, BatchNorm2d, Conv2d, GlobalAvgPool, bilinear_resize
from ..ops.concat import ChannelConcat
from .resnet import ResNet
from .xception import AlignedXception


class _ASPPBranch(nn.Module):
    def __init__(self, cin: int, cout: int, dilation: int):
        super().__init__()
        k, p = (1, 0) if dilation == 1 else (3, dilation)
        self.atrous_conv = Conv2d(cin, cout, k, padding=p, dilation=dilation, bias=False)
        self.bn = BatchNorm2d(cout)

    def forward(self, x, passthrough: bool = False, out=None, sole_consumer: bool = False):
        return conv_bn_act(self.atrous_conv, self.bn, x, relu=True, passthrough=passthrough, out=out,
                           sole_consumer=sole_consumer)


class _ImagePool(nn.Module):
    """global average pool → 1×1 conv → BN → ReLU with ``global_avg_pool.{0,1,2}`` key names."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.add_module("0", GlobalAvgPool())
        self.add_module("1", Conv2d(cin, cout, 1, bias=False))
        self.add_module("2", BatchNorm2d(cout))

    def forward(self, x):
        pooled = getattr(self, "0")(x)
        return conv_bn_act(getattr(self, "1"), getattr(self, "2"), pooled, relu=True)


class ASPP(nn.Module):
    def __init__(self, cin: int, output_stride: int, cout: int = 256, dropout: float = 0.5):
        super().__init__()
        rates = (1, 6, 12, 18) if output_stride == 16 else (1, 12, 24, 36)
        self.aspp1 = _ASPPBranch(cin, cout, rates[0])
        self.aspp2 = _ASPPBranch(cin, cout, rates[1])
        self.aspp3 = _ASPPBranch(cin, cout, rates[2])
        self.aspp4 = _ASPPBranch(cin, cout, rates[3])
        self.global_avg_pool = _ImagePool(cin, cout)
        self.conv1 = Conv2d(5 * cout, cout, 1, bias=False)
        self.bn1 = BatchNorm2d(cout)
        self.p_drop = dropout

    def forward(self, x):
        h, w = x.shape[2:]
        # five consumers of x: thread it through the four convolutions' autograd nodes so their
        # data-gradient kernels accumulate dL/dx in the store loop (no 4 add kernels over the
        # 2048-channel map; see models.resnet._FanOutMixin)
        # the branches write their channel slices of the concat buffer in place (ops.concat): no cat kernel
        cout = self.conv1.in_channels // 5
        cc = ChannelConcat(x, (cout,) * 5, (h, w))
        b1, x = self.aspp1(x, passthrough=True, out=cc.slot(0), sole_consumer=True)   # the chain's head: whole dL/dx
        b2, x = self.aspp2(x, passthrough=True, out=cc.slot(1))
        b3, x = self.aspp3(x, passthrough=True, out=cc.slot(2))
        b4, x = self.aspp4(x, passthrough=True, out=cc.slot(3))
        pooled = cc.broadcast(4, self.global_avg_pool(x))   # align_corners bilinear of a 1×1 map = broadcast
        y = cc.result(b1, b2, b3, b4, pooled)
        return conv_bn_act(self.conv1, self.bn1, y, relu=True, dropout=self.p_drop)


class _LastConv(nn.Module):
    """conv3×3-BN-ReLU-Drop(.5)-conv3×3-BN-ReLU-Drop(.1)-conv1×1 with Sequential-style indices."""

    def __init__(self, cin: int, num_classes: int, mid: int = 256):
        super().__init__()
        self.add_module("0", Conv2d(cin, mid, 3, padding=1, bias=False))
        self.add_module("1", BatchNorm2d(mid))
        self.add_module("4", Conv2d(mid, mid, 3, padding=1, bias=False))
        self.add_module("5", BatchNorm2d(mid))
        self.add_module("8", Conv2d(mid, num_classes, 1))
        self.p_drop = (0.5, 0.1)          # (tests set it to zeros to compare against a reference without its own RNG)

    def forward(self, x, training: bool):
        # conv-BN-ReLU-Dropout: the dropout runs inside the BN kernels (ops.batchnorm, csrc/bn_relu.hip)
        x = conv_bn_act(getattr(self, "0"), getattr(self, "1"), x, relu=True, dropout=self.p_drop[0])
        # each BN+ReLU+Dropout output has ONE consumer: its BatchNorm backward prologue (mask, dropout scale, the two
        # reductions) runs inside that consumer's data-gradient kernel
        x = conv_bn_act(getattr(self, "4"), getattr(self, "5"), x, relu=True, dropout=self.p_drop[1],
                        sole_consumer=True)
        return getattr(self, "8")(x, sole_consumer=True)


class Decoder(nn.Module):
    def __init__(self, low_level_channels: int, num_classes: int):
        super().__init__()
        self.conv1 = Conv2d(low_level_channels, 48, 1, bias=False)
        self.bn1 = BatchNorm2d(48)
        self.last_conv = _LastConv(256 + 48, num_classes)

    def forward(self, x, low):
        size = low.shape[2:]
        cc = ChannelConcat(x, (x.shape[1], self.conv1.out_channels), size)
        if tuple(x.shape[2:]) == tuple(size):
            cc.in_place = False                       # nothing to resize: no producer to write slot 0
        low = conv_bn_act(self.conv1, self.bn1, low, relu=True, out=cc.slot(1))
        x = bilinear_resize(x, size, out=cc.slot(0))
        return self.last_conv(cc.result(x, low), self.training)


class DeepLabV3Plus(nn.Module):
    def __init__(self, num_classes: int = 21, backbone: str = "resnet101", output_stride: int = 16,
                 in_channels: int = 3):
        super().__init__()
        if backbone.startswith("resnet"):
            self.backbone = ResNet(backbone, output_stride, in_channels)
        elif backbone in ("xception", "xception65"):
            self.backbone = AlignedXception(output_stride, in_channels)
        else:
            raise ValueError(f"unknown backbone {backbone}")
        self.aspp = ASPP(self.backbone.out_channels, output_stride)
        self.decoder = Decoder(self.backbone.low_level_channels, num_classes)
        self.num_classes = num_classes
        # When set (by the training engine, for a criterion that upsamples inside the loss kernel:
        # ops.loss / csrc/upsample_ce.hip) the training forward returns the stride-4 logits and the
        # 513² map is never written; evaluation always returns full-resolution logits.
        self.defer_upsample = False

    def forward(self, x):
        size = x.shape[2:]
        feat, low = self.backbone(x)
        y = self.decoder(self.aspp(feat), low)
        if self.defer_upsample and self.training:
            return y
        return bilinear_resize(y, size)

    # parameter groups named as in /root/reference/train_pascal.py:90-91
    def get_1x_lr_params(self) -> Iterator[nn.Parameter]:
        return (p for p in self.backbone.parameters() if p.requires_grad)

    def get_10x_lr_params(self) -> Iterator[nn.Parameter]:
        for m in (self.aspp, self.decoder):
            for p in m.parameters():
                if p.requires_grad:
                    yield p
