"""``LR_Scheduler('poly', lr, epochs, iters_per_epoch)`` called as
``scheduler(optimizer, i, epoch, best)`` — the interface the reference imports from an external
package and then leaves commented out (`/root/reference/train_pascal.py:34`, `:164`, `:223`), so
its effective schedule is constant LR.  Modes: ``poly`` (DeepLab, power 0.9), ``cos``, ``step``,
``const``; optional linear warm-up.  Group 0 gets ``lr``; further groups keep their ratio to
group 0 (the 1×/10× head multiplier hinted at `:90-91`)."""
from __future__ import annotations

import math


class LR_Scheduler:
    def __init__(self, mode: str, base_lr: float, num_epochs: int, iters_per_epoch: int,
                 lr_step: int = 0, warmup_epochs: float = 0.0, power: float = 0.9, min_lr: float = 0.0):
        if mode not in ("poly", "cos", "step", "const"):
            raise ValueError(f"unknown lr mode {mode}")
        if mode == "step" and lr_step <= 0:
            raise ValueError("step mode needs lr_step > 0")
        self.mode, self.base_lr, self.power, self.min_lr = mode, base_lr, power, min_lr
        self.iters_per_epoch = max(int(iters_per_epoch), 1)
        self.total = max(int(num_epochs) * self.iters_per_epoch, 1)
        self.lr_step = lr_step
        self.warmup = int(warmup_epochs * self.iters_per_epoch)
        self.epoch = -1
        self.current = base_lr

    def lr_at(self, it: int, epoch: int) -> float:
        t = min(epoch * self.iters_per_epoch + it, self.total)
        if self.mode == "poly":
            lr = self.base_lr * (1.0 - t / self.total) ** self.power
        elif self.mode == "cos":
            lr = 0.5 * self.base_lr * (1.0 + math.cos(math.pi * t / self.total))
        elif self.mode == "step":
            lr = self.base_lr * (0.1 ** (epoch // self.lr_step))
        else:
            lr = self.base_lr
        if self.warmup > 0 and t < self.warmup:
            lr = lr * (t + 1) / self.warmup
        return max(lr, self.min_lr)

    def __call__(self, optimizer, i: int, epoch: int, best_pred: float = 0.0) -> float:
        lr = self.lr_at(i, epoch)
        if epoch > self.epoch:
            self.epoch = epoch
        groups = optimizer.param_groups
        base0 = groups[0].get("initial_lr", self.base_lr) or self.base_lr
        for pg in groups:
            ratio = pg.get("initial_lr", base0) / base0 if base0 else 1.0
            pg["lr"] = lr * ratio
        self.current = lr
        return lr
