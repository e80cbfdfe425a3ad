"""Operator layer: hand-written gfx950 kernels behind autograd wrappers."""
from . import workspace  # noqa: F401
from ._backend import (NativeLibraryMissing, backend, deterministic, deterministic_mode, get_backend,  # noqa: F401
                       lib_built, set_backend, set_deterministic, use_hip)
from .augment import (IMAGENET_MEAN, IMAGENET_STD, affine_augment, affine_augment2, bbox_mask,  # noqa: F401
                      compose_crop, confidence_map, dynamic_crop_relax, expand_bbox, extreme_points,
                      extreme_points_heatmap, instance_augment, make_affine_params, make_warp_params, mask_bbox,
                      nellipse_guidance_, paste_jaccard)
from .batchnorm import (BatchNorm2d, batch_norm_act, convert_sync_batchnorm,  # noqa: F401
                        freeze_batchnorm)
from .bnfold import conv_bn_add_relu  # noqa: F401
from .conv import Conv2d, conv2d, conv_bn_act, conv_out_size  # noqa: F401
from .loss import SoftmaxCrossEntropy, softmax_cross_entropy  # noqa: F401
from .metrics import confusion_matrix, miou_from_confusion  # noqa: F401
from .pool import GlobalAvgPool, MaxPool3x3s2, global_avg_pool, max_pool3x3s2  # noqa: F401
from .sgd import sgd_flat_step_  # noqa: F401
from .upsample import Upsample, bilinear_resize  # noqa: F401
