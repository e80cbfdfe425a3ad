"""Dataset, sample-dict transforms and helpers (SURVEY.md layers L0–L2)."""
from . import helpers, imgproc, nellipse, skewed_axes_weight_map, transforms  # noqa: F401
from .combine_dbs import CombineDBs  # noqa: F401
from .pascal import VOC_CLASSES, SyntheticVOC, VOCSegmentation, write_synthetic_voc_tree  # noqa: F401
from .transforms import Compose  # noqa: F401
