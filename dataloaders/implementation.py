"""``from dataloaders.implementation import *`` (`/root/reference/train_pascal.py:36`): evaluation
helpers used by the script body — ``crop2fullmask``, ``calc_jaccard``, ``colorMaskWithAlpha``,
``generate_param_report``."""
from distributedpytorch_amd.data.helpers import (calc_jaccard, colorMaskWithAlpha, crop2fullmask,  # noqa: F401
                                                 generate_param_report, get_bbox, tens2image)
