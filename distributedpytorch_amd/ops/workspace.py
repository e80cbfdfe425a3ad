"""Per-step zero-initialised fp32 scratch pool.

The reduction kernels (fused BN statistics in the conv epilogue, BN backward sums, depthwise
weight gradients) accumulate with atomics into small zeroed buffers.  Allocating each with
``torch.zeros`` costs one fill launch per layer (~350 per training step).  Instead one pooled
device buffer is zeroed ONCE per step (a single memset on the stream) and sliced: every slice is
handed out at most once between two ``begin_step()`` calls, so it is guaranteed to still be zero.
If the pool runs dry or was never armed (no ``begin_step`` caller, e.g. unit tests) it falls back to ``torch.zeros``.
"""
from __future__ import annotations

from typing import Dict

import torch

_POOL_FLOATS = 64 << 20   # 256 MiB of the 288 GB (the folded separable units' weight-gradient scratch: 63 x 2.4 MB per
                          # Xception-65 step); only what a step handed out is zeroed again
_ALIGN = 64


class _Pool:
    def __init__(self, device: torch.device):
        self.buf = torch.zeros(_POOL_FLOATS, dtype=torch.float32, device=device)
        self.cursor = 0
        self.high = 0
        self.armed = False   # only hand out slices after begin_step() zeroed them

    def begin(self) -> None:
        if self.high > 0:
            self.buf[: self.high].zero_()
        self.cursor = 0
        self.high = 0
        self.armed = True

    def take(self, n: int):
        if not self.armed:
            return None
        n_al = (n + _ALIGN - 1) // _ALIGN * _ALIGN
        if self.cursor + n_al > _POOL_FLOATS:
            return None
        out = self.buf[self.cursor:self.cursor + n]
        self.cursor += n_al
        self.high = max(self.high, self.cursor)
        return out


_pools: Dict[torch.device, _Pool] = {}


def begin_step(device: torch.device) -> None:
    """Zero everything handed out since the previous call and rewind (one memset)."""
    if device.type != "cuda":
        return
    pool = _pools.get(device)
    if pool is None:
        pool = _pools[device] = _Pool(device)
    pool.begin()


def zeros(n: int, device: torch.device) -> torch.Tensor:
    pool = _pools.get(device)
    if pool is not None:
        t = pool.take(n)
        if t is not None:
            return t
    return torch.zeros(n, dtype=torch.float32, device=device)
