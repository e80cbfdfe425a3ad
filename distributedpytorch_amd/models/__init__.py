"""Model zoo + ``build_model`` factory (the reference's TODO "# build_model",
`/root/reference/train_pascal.py:6`)."""
from __future__ import annotations

import torch.nn as nn

from ..ops import convert_sync_batchnorm, freeze_batchnorm
from .danet import DANet
from .deeplab import ASPP, Decoder, DeepLabV3Plus
from .fcn import FCN
from .resnet import ResNet
from .xception import AlignedXception

MODEL_NAMES = (
    "fcn_resnet18", "fcn_resnet50", "fcn_resnet101",
    "deeplabv3plus_resnet18", "deeplabv3plus_resnet50", "deeplabv3plus_resnet101",
    "deeplabv3plus_xception65", "danet_resnet50", "danet_resnet101",
)

_ALIASES = {
    "deeplabv3plus_r101": "deeplabv3plus_resnet101",
    "deeplabv3plus_r50": "deeplabv3plus_resnet50",
    "deeplabv3plus_x65": "deeplabv3plus_xception65",
    "deeplabv3plus_xception": "deeplabv3plus_xception65",
    "fcn_r18": "fcn_resnet18",
    "danet": "danet_resnet101",
}


def build_model(name: str = "deeplabv3plus_resnet101", num_classes: int = 21, output_stride: int | None = None,
                sync_bn: bool = False, freeze_bn: bool = False, in_channels: int = 3,
                process_group=None) -> nn.Module:
    """Construct a segmentation network by name.

    ``sync_bn`` / ``freeze_bn`` mirror the knobs of the reference's commented DeepLab constructor
    (`/root/reference/train_pascal.py:85`)."""
    name = _ALIASES.get(name, name)
    head, _, backbone = name.partition("_")
    if head == "fcn":
        model = FCN(num_classes, backbone, output_stride or 32, in_channels)
    elif head == "deeplabv3plus":
        model = DeepLabV3Plus(num_classes, backbone, output_stride or 16, in_channels)
    elif head == "danet":
        model = DANet(num_classes, backbone, in_channels, output_stride or 8)
    else:
        raise ValueError(f"unknown model '{name}'; choose from {MODEL_NAMES}")
    if sync_bn:
        convert_sync_batchnorm(model, process_group)
    if freeze_bn:
        freeze_batchnorm(model)
    return model


__all__ = ["build_model", "MODEL_NAMES", "DeepLabV3Plus", "FCN", "DANet", "ResNet", "AlignedXception",
           "ASPP", "Decoder"]
