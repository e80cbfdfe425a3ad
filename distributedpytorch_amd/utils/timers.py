"""Per-phase step timers.  The reference only has a per-epoch wall-clock print
(`/root/reference/train_pascal.py:181`, `:307-308`).  On GPU phases are bracketed with HIP events
(no host sync inside the step; the events are resolved when ``summary()`` is read), on CPU with
``perf_counter``.  ``roctx`` ranges are emitted when the roctx library is loadable so ``rocprofv3
--marker-trace`` can attribute kernels to phases."""
from __future__ import annotations

import ctypes
import time
from collections import defaultdict
from contextlib import contextmanager
from typing import Dict, List

import torch

_roctx = None


def _load_roctx():
    global _roctx
    if _roctx is None:
        try:
            _roctx = ctypes.CDLL("librocprofiler-sdk-roctx.so")
        except OSError:
            try:
                _roctx = ctypes.CDLL("libroctx64.so")
            except OSError:
                _roctx = False
    return _roctx


class PhaseTimer:
    def __init__(self, use_gpu: bool, markers: bool = False):
        self.use_gpu = use_gpu and torch.cuda.is_available()
        self.markers = markers and bool(_load_roctx())
        self._events: Dict[str, List] = defaultdict(list)
        self._cpu: Dict[str, List[float]] = defaultdict(list)

    @contextmanager
    def phase(self, name: str):
        if self.markers:
            _roctx.roctxRangePushA(name.encode())
        if self.use_gpu:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            try:
                yield
            finally:
                b.record()
                self._events[name].append((a, b))
        else:
            t = time.perf_counter()
            try:
                yield
            finally:
                self._cpu[name].append((time.perf_counter() - t) * 1e3)
        if self.markers:
            _roctx.roctxRangePop()

    def summary(self, reset: bool = True) -> Dict[str, float]:
        """Mean milliseconds per phase since the last reset."""
        out: Dict[str, float] = {}
        if self.use_gpu:
            torch.cuda.synchronize()
            for k, evs in self._events.items():
                out[k] = sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1)
        for k, ts in self._cpu.items():
            out[k] = sum(ts) / max(len(ts), 1)
        if reset:
            self._events.clear()
            self._cpu.clear()
        return out
