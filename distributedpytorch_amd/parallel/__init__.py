"""One-process-per-GPU runtime: rendezvous, arenas, bucketed all-reduce DDP, sampler, engine."""
from . import dist_env  # noqa: F401
from .arena import ParamArena  # noqa: F401
from .ddp import DistributedDataParallel  # noqa: F401
from .dist_env import DistInfo, init_distributed  # noqa: F401
from .engine import TrainEngine, named_lr_groups  # noqa: F401
from .graphs import GraphedForward  # noqa: F401
from .optim import FlatSGD  # noqa: F401
from .sampler import DistributedSampler  # noqa: F401
