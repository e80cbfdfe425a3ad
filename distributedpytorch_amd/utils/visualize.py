"""Validation figure: the reference's 4-panel plot of the first validation batch — input with the
ground truth overlaid, the sigmoid prediction, and the position- and channel-attention head outputs
(`/root/reference/train_pascal.py:257-278`).  The reference pushes it to a hosted experiment
tracker; here it is a PNG in the run directory (no external service)."""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch

from ..data import helpers


def _to_np(t) -> np.ndarray:
    if torch.is_tensor(t):
        t = t.detach().float().cpu().numpy()
    return np.asarray(t)


def save_validation_panels(path: str, image_chw, gt_hw, outputs: Sequence, epoch: int,
                           semantic: bool = False, ignore_index: int = 255) -> Optional[str]:
    """``image_chw`` [≥3,H,W] in 0..255 (RGB first), ``gt_hw`` [H,W], ``outputs``: the model's output tuple for
    that sample ([C,h,w] each; a single tensor is accepted).  Instance task: panels 2–4 are sigmoid(out[0]),
    out[1] (position attention), out[2] (channel attention) as in the reference; semantic task: arg-max class
    maps.  Returns the path written (None when matplotlib is unavailable)."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:          # pragma: no cover - matplotlib is part of the image
        return None
    if torch.is_tensor(outputs):
        outputs = (outputs,)
    img = np.transpose(_to_np(image_chw)[:3], (1, 2, 0))
    if img.min() < 0:           # normalised input (semantic task): bring back to a displayable range
        img = (img - img.min()) / max(float(img.max() - img.min()), 1e-6) * 255.0
    gt = _to_np(gt_hw).astype(np.float32)
    titles = ["prediction", "Position Att", "Channel Att"]
    panels = []
    for k in range(3):
        if k < len(outputs) and outputs[k] is not None:
            o = _to_np(outputs[k])
            if semantic:
                panels.append(o.argmax(0).astype(np.float32) if o.ndim == 3 else o)
            else:
                o = o[0] if o.ndim == 3 else o
                panels.append(1.0 / (1.0 + np.exp(-np.clip(o, -60.0, 60.0))) if k == 0 else o)
        else:
            panels.append(None)
    fig = plt.figure(figsize=(16, 4.4))
    ax = fig.add_subplot(141)
    ax.imshow(np.clip(img, 0, 255).astype(np.uint8))
    mask = (gt == 1).astype(np.float32) if not semantic else ((gt > 0) & (gt != ignore_index)).astype(np.float32)
    if mask.shape == img.shape[:2]:
        ax.imshow(helpers.colorMaskWithAlpha(mask, color="r", transparency=0.3))
    ax.set_title("input + ground truth")
    ax.axis("off")
    for k, pnl in enumerate(panels):
        ax = fig.add_subplot(1, 4, k + 2)
        if pnl is not None:
            ax.imshow(pnl)
        ax.set_title(titles[k] if pnl is not None else titles[k] + " (n/a)")
        ax.axis("off")
    fig.suptitle("Epoch {0}".format(epoch))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    fig.savefig(path, dpi=80, bbox_inches="tight")
    plt.close(fig)
    return path
