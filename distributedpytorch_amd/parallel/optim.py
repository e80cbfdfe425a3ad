"""Arena-backed SGD with ``torch.optim``-style ``param_groups`` (so LR schedulers and the
reference's ``optimizer.param_groups[0]['lr']`` print, `/root/reference/train_pascal.py:182`, work).

``step()`` = one fused kernel per contiguous arena segment (ops.sgd, K10); it also applies the DDP
``1/world`` gradient scale, refreshes the bf16 compute copy and zeroes the gradients, i.e. it
subsumes ``optimizer.step(); optimizer.zero_grad()`` (`train_pascal.py:222-224`).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from ..ops.sgd import sgd_flat_step_
from .arena import ParamArena


class FlatSGD:
    def __init__(self, arena: ParamArena, lr: float | Sequence[float], momentum: float = 0.9,
                 weight_decay: float = 5e-4):
        ngroups = max(s.group for s in arena.slots) + 1
        lrs = [lr] * ngroups if isinstance(lr, (int, float)) else list(lr)
        assert len(lrs) == ngroups
        self.arena = arena
        self.param_groups: List[Dict] = [
            {"lr": float(lrs[g]), "initial_lr": float(lrs[g]), "momentum": momentum,
             "weight_decay": weight_decay,
             "params": [s.param for s in arena.slots if s.group == g]}
            for g in range(ngroups)
        ]
        self._steps = 0

    def __str__(self) -> str:
        g = self.param_groups[0]
        return (f"FlatSGD(lr={[pg['lr'] for pg in self.param_groups]}, momentum={g['momentum']}, "
                f"weight_decay={g['weight_decay']}, {self.arena.describe()})")

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        a = self.arena
        for (st, en, g) in a.segments:
            pg = self.param_groups[g]
            lowp = a.flat_lowp[st:en] if a.flat_lowp is not None else None
            sgd_flat_step_(a.flat_p[st:en], a.flat_g[st:en], a.flat_m[st:en], lowp,
                           pg["lr"], pg["momentum"], pg["weight_decay"], grad_scale,
                           zero_grad, first_step=(self._steps == 0))
        self._steps += 1
        a.refresh_transposed()      # dgrad weights of every conv, one launch (no-op off the HIP path)
        if a.flat_lowp is not None and not a.flat_p.is_cuda:
            for s in a.slots:  # CPU path updated masters through torch ops (version bump)
                s.param._dpamd_lowp_version = s.param._version

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.arena.zero_grad()

    # ---- sidecar state (optimiser state lives beside, not inside, the weights) ----
    def state_dict(self) -> Dict:
        return {"momentum": self.arena.flat_m.detach().cpu(), "steps": self._steps,
                "names": [s.name for s in self.arena.slots],
                "offsets": [s.offset for s in self.arena.slots],
                "lrs": [pg["lr"] for pg in self.param_groups]}

    def load_state_dict(self, sd: Dict) -> None:
        if sd["names"] != [s.name for s in self.arena.slots]:
            raise ValueError("optimizer sidecar does not match this model's parameter arena")
        self.arena.flat_m.copy_(sd["momentum"].to(self.arena.flat_m.device))
        self._steps = int(sd["steps"])
        for pg, lr in zip(self.param_groups, sd["lrs"]):
            pg["lr"] = lr
