"""Auxiliary HIP stream for work that is off the critical path of backward.

In backward every convolution produces two results from the same ``dy``: the data gradient, which
the NEXT backward node needs, and the weight gradient, which nobody needs before the gradient
all-reduce / optimiser.  Launched on one stream they serialise, and the layer-3/4 launches of
DeepLabv3+ R101 (tens of µs, a few hundred workgroups) leave most of the 256 CUs idle in their tail
rounds.  The weight-gradient kernels therefore go to a second stream (its own hardware queue), where
they fill the CUs the dgrad → BN-backward chain leaves free.

Ordering contract
* ``fork()``    — aux waits for everything enqueued on the current stream so far (``dy`` is ready);
* tensors read on aux are ``record_stream``-ed so the caching allocator does not recycle them early;
* ``join()``    — the current stream waits for everything enqueued on aux (call before the gradients
  are consumed: bucket all-reduce, optimiser step).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

_ENABLED = os.environ.get("DPAMD_AUX_STREAM", "1") != "0"
_streams: Dict[int, "torch.cuda.Stream"] = {}
_dirty: Dict[int, bool] = {}


def enabled() -> bool:
    return _ENABLED


def set_enabled(flag: bool) -> None:
    global _ENABLED
    _ENABLED = bool(flag)


def aux_stream(device: Optional[torch.device] = None) -> "torch.cuda.Stream":
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    st = _streams.get(idx)
    if st is None:
        st = _streams[idx] = torch.cuda.Stream(device=idx)
    return st


def fork(device: Optional[torch.device] = None) -> "torch.cuda.Stream":
    """Make the aux stream wait for the current stream's work so far and return it."""
    st = aux_stream(device)
    st.wait_stream(torch.cuda.current_stream(st.device))
    _dirty[st.device.index] = True
    return st


def join(device: Optional[torch.device] = None) -> None:
    """Make the current stream wait for all work enqueued on the aux stream."""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    if _dirty.get(idx):
        torch.cuda.current_stream(idx).wait_stream(_streams[idx])
        _dirty[idx] = False
