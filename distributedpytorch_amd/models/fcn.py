"""FCN head on a ResNet backbone — the CPU/gloo plumbing model of ``BASELINE.json`` config 1
(FCN ResNet-18, synthetic 256×256, world_size 2)."""
from __future__ import annotations

import torch.nn as nn
import torch.nn.functional as F

from ..ops import conv_bn_act, BatchNorm2d, Conv2d, bilinear_resize
from .resnet import ResNet


class FCNHead(nn.Module):
    def __init__(self, cin: int, num_classes: int):
        super().__init__()
        mid = max(cin // 4, 16)
        self.conv = Conv2d(cin, mid, 3, padding=1, bias=False)
        self.bn = BatchNorm2d(mid)
        self.cls = Conv2d(mid, num_classes, 1)

    def forward(self, x, training: bool):
        x = conv_bn_act(self.conv, self.bn, x, relu=True)
        return self.cls(F.dropout(x, 0.1, training))


class FCN(nn.Module):
    def __init__(self, num_classes: int = 21, backbone: str = "resnet18", output_stride: int = 32,
                 in_channels: int = 3):
        super().__init__()
        self.backbone = ResNet(backbone, output_stride, in_channels, multi_grid=None)
        self.head = FCNHead(self.backbone.out_channels, num_classes)
        self.num_classes = num_classes
        self.defer_upsample = False

    def forward(self, x):
        feat, _ = self.backbone(x)
        y = self.head(feat, self.training)
        if self.defer_upsample and self.training:
            return y      # the criterion upsamples inside the loss kernel (see DeepLabV3Plus)
        return bilinear_resize(y, x.shape[2:])

    def get_1x_lr_params(self):
        return (p for p in self.backbone.parameters() if p.requires_grad)

    def get_10x_lr_params(self):
        return (p for p in self.head.parameters() if p.requires_grad)
