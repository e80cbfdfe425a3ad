"""Arena-backed SGD with ``torch.optim``-style ``param_groups`` (so LR schedulers and the
reference's ``optimizer.param_groups[0]['lr']`` print, `/root/reference/train_pascal.py:182`, work).

``step()`` = one fused kernel per contiguous arena segment (ops.sgd, K10); it also applies the DDP
``1/world`` gradient scale, refreshes the bf16 compute copy and zeroes the gradients, i.e. it
subsumes ``optimizer.step(); optimizer.zero_grad()`` (`train_pascal.py:222-224`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..ops.sgd import sgd_flat_step_
from .arena import ParamArena, _view_like


class FlatSGD:
    def __init__(self, arena: ParamArena, lr: float | Sequence[float], momentum: float = 0.9,
                 weight_decay: float = 5e-4, adapters: Optional[Sequence[Tuple[str, object]]] = None):
        """``adapters``: ``(prefix, module)`` pairs whose ``state_to_logical(key, t)`` / ``state_to_physical(key, t)``
        translate per-parameter state of parameters named ``prefix + key`` between the stored layout and the
        architecture's (models.xception: the 728-channel flow stored 768 wide).  The sidecar then holds LOGICAL
        shapes, so a checkpoint written by a padded run resumes into an unpadded model and back."""
        self.adapters = list(adapters or [])
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
        # set by a capturing TrainEngine (graph_step): fp32 [groups] on the device — the SGD launches of a captured
        # step read their learning rate from here instead of from the (frozen) kernel argument
        self.lr_dev: Optional[torch.Tensor] = None

    def __str__(self) -> str:
        g = self.param_groups[0]
        return (f"FlatSGD(lr={[pg['lr'] for pg in self.param_groups]}, momentum={g['momentum']}, "
                f"weight_decay={g['weight_decay']}, {self.arena.describe()})")

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        a = self.arena
        lib = None
        if self.lr_dev is not None:
            from ..ops import _backend
            lib = _backend.lib()
        try:
            for (st, en, g) in a.segments:
                pg = self.param_groups[g]
                lowp = a.flat_lowp[st:en] if a.flat_lowp is not None else None
                if lib is not None:
                    lib.dp_sgd_set_lr_ptr(self.lr_dev[g:g + 1].data_ptr())
                sgd_flat_step_(a.flat_p[st:en], a.flat_g[st:en], a.flat_m[st:en], lowp,
                               pg["lr"], pg["momentum"], pg["weight_decay"], grad_scale,
                               zero_grad, first_step=(self._steps == 0))
        finally:
            if lib is not None:
                lib.dp_sgd_set_lr_ptr(None)
        self._steps += 1
        a.refresh_transposed()      # dgrad weights of every conv, one launch (no-op off the HIP path)
        if a.flat_lowp is not None and not a.flat_p.is_cuda:
            for s in a.slots:  # CPU path updated masters through torch ops (version bump)
                s.param._dpamd_lowp_version = s.param._version

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.arena.zero_grad()

    # ---- sidecar state (optimiser state lives beside, not inside, the weights) ----
    def _adapt(self, name: str, t: torch.Tensor, to_logical: bool) -> torch.Tensor:
        for prefix, mod in self.adapters:
            if name.startswith(prefix):
                fn = mod.state_to_logical if to_logical else mod.state_to_physical
                return fn(name[len(prefix):], t)
        return t

    def state_dict(self) -> Dict:
        """Momentum PER PARAMETER in the architecture's shapes (not the arena's flat, possibly padded layout)."""
        from ..utils.hostcopy import to_host
        m = self.arena.flat_m.detach()
        # logical shapes (4-D weights lie KRSC in the arena); one device->host copy for all of them
        host = to_host([self._adapt(s.name, _view_like(m, s.offset, s.param), True) for s in self.arena.slots])
        by_name = {s.name: t for s, t in zip(self.arena.slots, host)}
        return {"format": 2, "momentum_by_name": by_name, "steps": self._steps,
                "lrs": [pg["lr"] for pg in self.param_groups]}

    def load_state_dict(self, sd: Dict) -> None:
        slots = self.arena.slots
        if "momentum_by_name" in sd:
            by_name = sd["momentum_by_name"]
            missing = [s.name for s in slots if s.name not in by_name]
            extra = sorted(set(by_name) - {s.name for s in slots})
            if missing or extra:
                raise ValueError(f"optimizer sidecar does not match this model: missing {missing[:3]}, "
                                 f"unexpected {extra[:3]}")
            dev = self.arena.flat_m.device
            for s in slots:
                t = self._adapt(s.name, by_name[s.name].to(dev), False)
                if tuple(t.shape) != tuple(s.param.shape):
                    raise ValueError(f"optimizer sidecar: momentum of '{s.name}' has shape {tuple(by_name[s.name].shape)}, "
                                     f"the parameter is {tuple(s.param.shape)}")
                _view_like(self.arena.flat_m, s.offset, s.param).copy_(t)
        else:
            # round-1/2 sidecars: one flat buffer in the WRITER's arena layout — only a reader with the identical
            # layout may take it (a padded Xception arena against an unpadded one would scramble the momentum)
            if sd.get("names") != [s.name for s in slots]:
                raise ValueError("optimizer sidecar does not match this model's parameter arena")
            if sd.get("offsets") != [s.offset for s in slots] or sd["momentum"].numel() != self.arena.flat_m.numel():
                raise ValueError("optimizer sidecar was written with another arena layout (e.g. the 768-wide stored "
                                 "Xception flow, DPAMD_X65_PAD); re-save it with this version or resume with the "
                                 "writer's padding setting")
            self.arena.flat_m.copy_(sd["momentum"].to(self.arena.flat_m.device))
        self._steps = int(sd["steps"])
        for pg, lr in zip(self.param_groups, sd["lrs"]):
            pg["lr"] = lr
