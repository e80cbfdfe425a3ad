"""hipGraph capture of launch-bound inner loops.

The reference evaluates with ``testBatch = 1`` (`/root/reference/train_pascal.py:60`, loop `:237-254`):
one 513² image per forward.  DeepLabv3+ R101 is ~350 kernel launches per forward and at batch 1 each
runs for a few microseconds, so the eager loop is bound by host launch cost, not by the GPU.  A
captured graph replays the whole forward as one submission.

``GraphedForward`` owns static input/output buffers; the captured region contains only kernels and
allocator-private memory, so everything data-dependent on the host (autotuning, lazy bf16 weight
casts) is flushed by eager warm-up runs before capture.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn as nn


class GraphedForward:
    """Inference forward of ``model`` (eval mode, no grad) as a hipGraph, one graph per input shape."""

    def __init__(self, model: nn.Module, warmup: int = 2):
        self.model = model
        self.warmup = warmup
        self._graphs: Dict[Tuple, tuple] = {}
        from ..ops.conv import Conv2d
        self._convs = [m for m in model.modules() if isinstance(m, Conv2d)]
        # modules with constants of the evaluation phase held in fixed storage (BatchNorm's scale | shift blocks, the
        # folded weights of Xception's separable units)
        self._consts = [m for m in model.modules() if hasattr(m, "eval_refresh")]

    def _capture(self, x: torch.Tensor):
        static_in = x.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):      # autotune / lazy casts happen here, outside the capture
                self.model(static_in)
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        graph = torch.cuda.CUDAGraph()
        # (thread_local: a loader's pin-memory thread copying next to the capture must not invalidate it)
        with torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
            static_out = self.model(static_in)
        return graph, static_in, static_out

    @torch.no_grad()
    def __call__(self, x: torch.Tensor):
        if self.model.training:
            raise RuntimeError("GraphedForward replays an inference graph: call model.eval() first")
        key = (tuple(x.shape), x.dtype, x.stride())
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._graphs[key] = self._capture(x)
        graph, static_in, static_out = entry
        # the bf16 weight copies are refreshed IN PLACE outside the graph when a master weight changed
        for m in self._convs:
            m.lowp_weight(x.dtype)
        for m in self._consts:     # same storage, refreshed when stale (ops.batchnorm, ops.sepconv)
            m.eval_refresh()
        static_in.copy_(x)
        graph.replay()
        return static_out      # valid until the next call with the same shape

    def reset(self) -> None:
        """Drop the captured graphs (after the weights' STORAGE changed, e.g. a re-built arena; plain
        weight updates need no reset: graphs hold pointers, not values)."""
        self._graphs.clear()
