"""Device confusion matrix for mIoU, SURVEY.md §2.5 K12 / §2.6 C6.

The reference accumulates per-object Jaccard on the host (`/root/reference/train_pascal.py:282-298`);
the north-star metric is 21-class mIoU, so predictions never leave the GPU: an LDS-atomic histogram
kernel builds the ``K×K`` matrix and one small all-reduce merges ranks."""
from __future__ import annotations

import torch

from . import _backend as B


@torch.no_grad()
def confusion_matrix(pred: torch.Tensor, target: torch.Tensor, num_classes: int,
                     ignore_index: int = 255, out: torch.Tensor | None = None) -> torch.Tensor:
    """``out[t, p] += 1`` for every pixel with ``target != ignore_index`` (and ``target < K``).

    ``pred``: integer class map [N,H,W] *or* logits [N,K,H,W] (arg-max taken on device)."""
    if out is None:
        out = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=pred.device)
    if B.use_hip(pred):
        from . import _hip_misc
        _hip_misc.confusion(pred, target, num_classes, ignore_index, out)
        return out
    if pred.dim() == 4:
        pred = pred.argmax(dim=1)
    t = target.reshape(-1).long()
    p = pred.reshape(-1).long()
    keep = (t != ignore_index) & (t >= 0) & (t < num_classes)
    idx = t[keep] * num_classes + p[keep]
    out += torch.bincount(idx, minlength=num_classes * num_classes).view(num_classes, num_classes)
    return out


def miou_from_confusion(cm: torch.Tensor):
    """Returns (mIoU over classes that appear, per-class IoU, pixel accuracy)."""
    cm = cm.double()
    tp = cm.diag()
    denom = cm.sum(0) + cm.sum(1) - tp
    present = denom > 0
    iou = torch.where(present, tp / denom.clamp_min(1), torch.zeros_like(tp))
    miou = iou[present].mean().item() if present.any() else 0.0
    acc = (tp.sum() / cm.sum().clamp_min(1)).item()
    return miou, iou, acc
