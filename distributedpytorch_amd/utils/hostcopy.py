"""Many device tensors -> host with ONE copy per dtype.

A checkpoint of DeepLabv3+ R101 is 680 state tensors + 341 momentum tensors; ``t.cpu()`` on each is a blocking
device->host copy of a few KiB (1,021 host-device synchronisations per checkpoint, `tools/find_syncs.py`).  Here the
tensors of one (device, dtype) are flattened into one device buffer, copied once, and handed back as independent
contiguous host tensors in their own (logical) shapes."""
from __future__ import annotations

from typing import List, Sequence

import torch


def to_host(tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    out: List[torch.Tensor] = [None] * len(tensors)          # type: ignore[list-item]
    groups: dict = {}
    for i, t in enumerate(tensors):
        t = t.detach()
        if t.device.type == "cpu":
            out[i] = t.contiguous().clone()
        else:
            groups.setdefault((t.device, t.dtype), []).append((i, t))
    for members in groups.values():
        flat = torch.cat([t.reshape(-1) for _, t in members]).cpu()      # reshape(-1): logical (NCHW) element order
        off = 0
        for i, t in members:
            n = t.numel()
            out[i] = flat[off:off + n].view(t.shape).clone()             # own storage: the file keeps one per tensor
            off += n
    return out
