"""Fused flat-buffer SGD (momentum + weight decay), SURVEY.md §2.5 K10.

Reference: ``optim.SGD(net.parameters(), lr, momentum, weight_decay)`` then
``optimizer.step(); optimizer.zero_grad()`` (`/root/reference/train_pascal.py:118`, `:222-224`).
Here the parameters, gradients, momentum and the bf16 compute copy live in flat arenas
(``parallel.arena``) and ONE kernel launch per segment performs

    g   = g * grad_scale + wd * p          (grad_scale folds the 1/world_size DDP average, C3)
    m   = mu * m + g
    p  -= lr * m
    p16 = bf16(p)                          (refreshes the compute copy: no separate cast kernels)
    g   = 0                                (optional: folds ``zero_grad``)

matching ``torch.optim.SGD`` (dampening 0, no Nesterov) bit-for-bit in fp32 up to fma contraction.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _backend as B


@torch.no_grad()
def sgd_flat_step_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, p_lowp: Optional[torch.Tensor],
                   lr: float, momentum: float, weight_decay: float, grad_scale: float = 1.0,
                   zero_grad: bool = True, first_step: bool = False) -> None:
    """In-place update of 1-D fp32 ``p``/``m`` from ``g``.  ``first_step`` seeds ``m = g`` like torch."""
    assert p.dim() == 1 and p.shape == g.shape == m.shape
    if B.use_hip(p):
        from . import _hip_misc
        _hip_misc.sgd_flat(p, g, m, p_lowp, lr, momentum, weight_decay, grad_scale, zero_grad, first_step)
        return
    if grad_scale != 1.0:
        g.mul_(grad_scale)
    if weight_decay != 0.0:
        g.add_(p, alpha=weight_decay)
    if momentum != 0.0:
        if first_step:
            m.copy_(g)
        else:
            m.mul_(momentum).add_(g)
        p.add_(m, alpha=-lr)
    else:
        p.add_(g, alpha=-lr)
    if p_lowp is not None:
        p_lowp.copy_(p)
    if zero_grad:
        g.zero_()
