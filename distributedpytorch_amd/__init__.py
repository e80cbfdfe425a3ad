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

__version__ = "0.1.0"

from . import ops  # noqa: F401
