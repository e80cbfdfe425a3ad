"""Top-level ``custom_transforms`` module (same name as `/root/reference/custom_transforms.py`);
implementation in ``distributedpytorch_amd.data.transforms``."""
from distributedpytorch_amd.data.transforms import *  # noqa: F401,F403
from distributedpytorch_amd.data.transforms import (AddConfidenceMap, Compose, ConcatInputs,  # noqa: F401
                                                    CreateBBMask, CropFromMask, CropFromMaskStatic,
                                                    ExtremePoints, FixedResize, FixedScaleCenterCrop,
                                                    NEllipse, NEllipseWithGaussians, Normalize,
                                                    RandomHorizontalFlip, RandomScaleCrop, ScaleNRotate,
                                                    SemSegToTensor, ToImage, ToTensor, trainValMode)
