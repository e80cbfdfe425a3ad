"""Checkpointing, LR schedule, losses, logging, timers."""
from .checkpoint import load_checkpoint, load_weights, reference_state_dict, save_checkpoint  # noqa: F401
from .logging import JsonlLogger  # noqa: F401
from .losses import SegmentationMultiLosses, class_balanced_cross_entropy_loss  # noqa: F401
from .lr_scheduler import LR_Scheduler  # noqa: F401
from .timers import PhaseTimer  # noqa: F401
