"""distributedpytorch_amd — MI355X-native distributed segmentation trainer.

Layer map (see SURVEY.md §1 for the reference's logical layers):

* ``data``      dataset (``VOCSegmentation``), sample-dict transforms, helpers, synthetic VOC
* ``ops``       hand-written HIP/CDNA4 kernels (gfx950) behind ``torch.autograd.Function`` wrappers
* ``models``    ResNet / Xception backbones, FCN / DeepLabv3+ / DANet heads, ``build_model``
* ``parallel``  one-process-per-GPU runtime: rendezvous, bucketed overlapped gradient all-reduce,
                SyncBatchNorm statistics all-reduce, distributed sampler, rank-0 guards
* ``utils``     checkpoints (reference-loadable ``module.``-prefixed state_dict), poly LR schedule,
                losses, mIoU / Jaccard metrics, JSONL logging, HIP-event timers
"""

__version__ = "0.2.0"

import os as _os

# dmabuf IPC is the only IPC mode of this platform's host driver; ROCr reads the variable once, at the first
# HIP call, so it has to be in the environment before anything initialises the GPU (parallel.dist_env checks).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import ops  # noqa: F401
