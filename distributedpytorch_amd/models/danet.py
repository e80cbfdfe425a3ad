"""DANet (dual attention) — the model the reference script actually instantiates:
``net = DANet(1, 'resnet101')`` (`/root/reference/train_pascal.py:32`, `:86`), imported from an
external package that is not in the reference tree.  Contract recovered from the
call sites: ``forward`` returns a 3-tuple ``(fused, position_branch, channel_branch)`` of logits at
input resolution (`:258-260` visualises ``[1]`` / ``[2]`` as the position / channel attention
outputs, ``[0]`` is the prediction used for Jaccard, `:283`).

The attention modules run fused hand-written kernels on the GPU (``ops.attention`` / csrc/attention.hip:
flash-style position attention that never materialises the (H·W)² map, gram + softmax + per-image 1×1
convolutions for the channel attention), like the convolutions, BN and resizes.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..ops import conv_bn_act, BatchNorm2d, Conv2d, bilinear_resize
from ..ops.attention import channel_attention, head_dropout, position_attention
from .resnet import ResNet


class PositionAttention(nn.Module):
    """Self-attention over the H·W positions of a feature map."""

    def __init__(self, channels: int):
        super().__init__()
        self.query_conv = Conv2d(channels, channels // 8, 1)
        self.key_conv = Conv2d(channels, channels // 8, 1)
        self.value_conv = Conv2d(channels, channels, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return position_attention(self.query_conv(x), self.key_conv(x), self.value_conv(x), x, self.gamma)


class ChannelAttention(nn.Module):
    """Self-attention over channels (no learned projections, only ``gamma``)."""

    def __init__(self):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return channel_attention(x, self.gamma)


class _ConvBNReLU(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("0", Conv2d(cin, cout, 3, padding=1, bias=False))
        self.add_module("1", BatchNorm2d(cout))

    def forward(self, x):
        return conv_bn_act(getattr(self, "0"), getattr(self, "1"), x, relu=True)


class _DropCls(nn.Module):
    def __init__(self, cin, nclass):
        super().__init__()
        self.add_module("1", Conv2d(cin, nclass, 1))

    def forward(self, x):              # x: already dropped (DANetHead.forward: one fused pass for the three branches)
        return getattr(self, "1")(x)


class DANetHead(nn.Module):
    def __init__(self, cin: int, nclass: int):
        super().__init__()
        mid = cin // 4
        self.conv5a, self.conv5c = _ConvBNReLU(cin, mid), _ConvBNReLU(cin, mid)
        self.sa, self.sc = PositionAttention(mid), ChannelAttention()
        self.conv51, self.conv52 = _ConvBNReLU(mid, mid), _ConvBNReLU(mid, mid)
        self.conv6, self.conv7, self.conv8 = _DropCls(mid, nclass), _DropCls(mid, nclass), _DropCls(mid, nclass)

    def forward(self, x, training: bool):
        sa = self.conv51(self.sa(self.conv5a(x)))
        sc = self.conv52(self.sc(self.conv5c(x)))
        d_sa, d_sc, d_sum = head_dropout(sa, sc, 0.1, training)
        return (self.conv8(d_sum), self.conv6(d_sa), self.conv7(d_sc))


class DANet(nn.Module):
    """``DANet(nclass, backbone)`` — same positional signature as the reference call site."""

    def __init__(self, nclass: int, backbone: str = "resnet101", in_channels: int = 3,
                 output_stride: int = 8):
        super().__init__()
        self.pretrained = ResNet(backbone, output_stride, in_channels)
        self.head = DANetHead(self.pretrained.out_channels, nclass)
        self.nclass = nclass

    def forward(self, x):
        size = x.shape[2:]
        feat, _ = self.pretrained(x)
        return tuple(bilinear_resize(o, size) for o in self.head(feat, self.training))

    def get_1x_lr_params(self):
        return (p for p in self.pretrained.parameters() if p.requires_grad)

    def get_10x_lr_params(self):
        return (p for p in self.head.parameters() if p.requires_grad)
