"""Debug-mode data sanity checks — the reference asserts on every batch, on the host, through
``.numpy()`` / ``np.unique`` (`/root/reference/train_pascal.py:188-190`, `:239-241`): RGB and guidance
channels within [0, 255] with more than two distinct values, ground truth exactly {0, 1}.  That is an
O(B·4·512²) host pass per step; here the same predicates are evaluated ON THE DEVICE (min / max and a
"more than two distinct values" test that needs no sort) and folded into ONE flag word that is read
only every ``every`` steps — enabled with ``DPAMD_DEBUG=1`` (off on the hot path)."""
from __future__ import annotations

import os

import torch

ENABLED = os.environ.get("DPAMD_DEBUG", "0") != "0"

_MSG = {1: "RGB channels outside [0, 255]", 2: "RGB channels have at most two distinct values",
        4: "guidance channel outside [0, 255]", 8: "guidance channel has at most two distinct values",
        16: "instance ground truth is not exactly {0, 1}", 32: "semantic labels outside [0, C) ∪ {ignore}",
        64: "non-finite input"}


def _more_than_two_values(t: torch.Tensor) -> torch.Tensor:
    lo, hi = t.min(), t.max()
    return ((t != lo) & (t != hi)).any()


class BatchSanity:
    def __init__(self, task: str, num_classes: int = 21, ignore_index: int = 255, normalised: bool = False,
                 every: int = 20):
        self.task, self.nc, self.ign, self.normalised, self.every = task, num_classes, ignore_index, normalised, every
        self.flags = None
        self.steps = 0

    @torch.no_grad()
    def check(self, inputs: torch.Tensor, target: torch.Tensor) -> None:
        dev = inputs.device
        if self.flags is None:
            self.flags = torch.zeros((), dtype=torch.int32, device=dev)
        f = torch.zeros((), dtype=torch.int32, device=dev)
        x = inputs.float()
        f |= (~torch.isfinite(x).all()).int() * 64
        rgb = x[:, :3]
        if not self.normalised:
            f |= ((rgb.max() > 255) | (rgb.min() < 0)).int() * 1
        f |= (~_more_than_two_values(rgb)).int() * 2
        if self.task == "instance":
            if x.shape[1] > 3:
                g = x[:, 3]
                f |= ((g.max() > 255) | (g.min() < 0)).int() * 4
                f |= (~_more_than_two_values(g)).int() * 8
            t = target.float()
            f |= (~((t.min() == 0) & (t.max() == 1) & ((t == 0) | (t == 1)).all())).int() * 16
        else:
            t = target.long()
            f |= (~(((t >= 0) & (t < self.nc)) | (t == self.ign)).all()).int() * 32
        self.flags |= f.to(self.flags.device)
        self.steps += 1
        if self.steps % self.every == 0:
            self.raise_if_flagged()

    def raise_if_flagged(self) -> None:
        if self.flags is None:
            return
        v = int(self.flags.item())
        if v:
            raise AssertionError("data sanity (DPAMD_DEBUG): " + "; ".join(m for b, m in _MSG.items() if v & b))
