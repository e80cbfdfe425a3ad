"""Auxiliary HIP stream for work that is off the critical path of backward.

In backward every convolution produces two results from the same ``dy``: the data gradient, which
the NEXT backward node needs, and the weight gradient, which nobody needs before the gradient
all-reduce / optimiser.  Launched on one stream they serialise, and the layer-3/4 launches of
DeepLabv3+ R101 (tens of µs, a few hundred workgroups) leave most of the 256 CUs idle in their tail
rounds.  The weight-gradient kernels therefore go to a second stream (its own hardware queue), where
they fill the CUs the dgrad → BN-backward chain leaves free.

Ordering contract
* ``fork()``    — aux waits for everything enqueued on the current stream so far (``dy`` is ready);
* tensors read on aux are handed to ``keep_alive()``: the section holds a reference until the current stream has
  waited for that section (a few forks later, or at ``join()``), so their memory returns to the allocator in stream
  order.  NOT ``record_stream``: that holds a block back until the GPU has passed an event recorded when the HOST
  released the tensor, and the host runs up to a whole backward ahead of the GPU — every block read by a weight
  gradient stayed unusable for most of the step and the allocator reserved twice what was allocated (R101 B256: 168
  vs 81 GiB; Xception-65 OS8 at B128 filled the device: profiles/r3_aux_stream_memory.md);
* ``join()``    — the current stream waits for everything enqueued on aux (call before the gradients
  are consumed: bucket all-reduce, optimiser step).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

_ENABLED = os.environ.get("DPAMD_AUX_STREAM", "1") != "0"
# Number of auxiliary streams used round-robin.  A weight-gradient kernel ends in a burst of fp32
# atomics (its split-K combine); the next kernel of the SAME stream cannot start until that burst has
# drained, a kernel of another stream can.  Measured at B32 (gpurun_out/s8_bench_b32_aux*.log): 697 img/s
# with one aux stream, 685 with two, so one is the default; GPU_MAX_HW_QUEUES is 4 (main + comm + 2 aux).
_NAUX = 1
# Queue priority of the auxiliary stream(s).  "normal" (default): a stream of PyTorch's pool.  "low": created with
# hipStreamCreateWithPriority at the LEAST priority of the device's range (torch.cuda.Stream only reaches the normal
# and high levels, hence the direct runtime call + torch.cuda.ExternalStream), so that when both hardware queues hold
# ready workgroups the dgrad -> BN-backward chain is dispatched first.  Low was the default of rounds 1-3 (+0.7 % at
# B128 in round 1; 0.0 / +0.3 / +0.9 % on R101 / Xception-65 / DANet at the end of round 3, same-box A/B) until a
# reproducible STALL was found with it: in a process that had created several other streams (device sample stores'
# copy streams), a ResNet-18 step stopped with the step's stream waiting at join() for the low-priority queue, which
# made no progress for minutes; the same sequence with this stream at normal priority (pool stream, or the same raw
# handle path at priority 0) completes, and so does the low-priority run with serialised launches
# (profiles/r3_low_priority_stream_stall.md).  No ordering cycle exists between the two streams (every wait
# refers to work enqueued earlier), so the queue is starved below the runtime; a fraction of a percent is not worth
# a job that can stop.  set_priority("low") remains for experiments.
_PRIORITY = "normal"
_raw_handles: list = []     # keeps the hipStream_t of the external streams (never destroyed: process lifetime)


def _make_stream(idx: int) -> "torch.cuda.Stream":
    if _PRIORITY == "low":
        import ctypes
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            least, greatest = ctypes.c_int(0), ctypes.c_int(0)
            with torch.cuda.device(idx):
                if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) == 0 \
                        and least.value > greatest.value:
                    handle = ctypes.c_void_p()
                    # flags = 1: hipStreamNonBlocking (no implicit ordering against the null stream)
                    if hip.hipStreamCreateWithPriority(ctypes.byref(handle), ctypes.c_uint(1), least) == 0 \
                            and handle.value:
                        _raw_handles.append(handle)
                        return torch.cuda.ExternalStream(handle.value, device=idx)
        except OSError:
            pass
    return torch.cuda.Stream(device=idx)
_streams: Dict[int, list] = {}
_next: Dict[int, int] = {}
_dirty: Dict[int, bool] = {}


def enabled() -> bool:
    """Auxiliary stream in use?  Never in deterministic mode (ops._backend): one stream, one order."""
    from . import _backend
    return _ENABLED and not _backend.deterministic()


def set_priority(level: str) -> None:
    """'normal' (default) | 'low'.  Streams that exist already are joined and dropped; the next fork creates new
    ones.  parallel.p2p calls this with 'normal' when ranks SHARE a device (rehearsals): a peer's spinning one-shot
    all-reduce kernel on a normal-priority queue outranks low-priority weight gradients, which the matching
    all-reduce of this rank waits for — both then sit until the bounded spin times out."""
    global _PRIORITY
    if level not in ("low", "normal"):
        raise ValueError("aux stream priority: 'low' or 'normal'")
    if level != _PRIORITY:
        for idx in list(_streams):
            join(torch.device("cuda", idx))
        _streams.clear()
        _next.clear()
        _PRIORITY = level


def aux_stream(device: Optional[torch.device] = None) -> "torch.cuda.Stream":
    """The next auxiliary stream of ``device`` (round-robin over _NAUX streams)."""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    pool = _streams.get(idx)
    if pool is None:
        pool = _streams[idx] = [_make_stream(idx) for _ in range(_NAUX)]
        _next[idx] = 0
    k = _next[idx]
    _next[idx] = (k + 1) % len(pool)
    return pool[k]


# A forked section = the work queued on the aux stream between two fork() calls + the tensors it reads.  The current
# stream waits for a section _MAX_LAG forks after it was opened (a GPU-side wait that costs nothing: the weight
# gradients are shorter than the dgrad -> BN-backward chain between two forks; measured identical step times for 2..16)
# and only then are the section's tensors released on the host.
_MAX_LAG = 4
_sections: Dict[int, list] = {}          # device -> [[end event or None, [tensors]], ...], oldest first


def fork(device: Optional[torch.device] = None) -> "torch.cuda.Stream":
    """Make the aux stream wait for the current stream's work so far and return it."""
    st = aux_stream(device)
    idx = st.device.index
    cur = torch.cuda.current_stream(st.device)
    q = _sections.setdefault(idx, [])
    if q and q[-1][0] is None:
        ev = torch.cuda.Event()
        ev.record(st)                    # the end of the section opened by the previous fork
        q[-1][0] = ev
    while len(q) > _MAX_LAG:
        ev, keep = q.pop(0)
        cur.wait_event(ev)
        keep.clear()                     # released behind the wait: reusable by the current stream in stream order
    q.append([None, []])
    st.wait_stream(cur)
    _dirty[idx] = True
    return st


def keep_alive(*tensors: Optional[torch.Tensor]) -> None:
    """The tensors are read by the section opened by the last ``fork()`` of their device: hold them until the current
    stream has waited for that section (see the module docstring)."""
    for t in tensors:
        if t is not None and t.is_cuda:
            q = _sections.get(t.device.index)
            if not q:
                raise RuntimeError("streams.keep_alive outside a forked section")
            q[-1][1].append(t)


def join(device: Optional[torch.device] = None) -> None:
    """Make the current stream wait for all work enqueued on the aux stream."""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    if _dirty.get(idx):
        cur = torch.cuda.current_stream(idx)
        for st in _streams[idx]:
            cur.wait_stream(st)
        _dirty[idx] = False
    q = _sections.pop(idx, None)         # everything forked so far is behind the current stream now
    if q:
        for _ev, keep in q:
            keep.clear()


def join_into(stream: "torch.cuda.Stream", device: Optional[torch.device] = None) -> None:
    """Make ``stream`` (not the current one) wait for all work enqueued on the aux stream(s) so far.  The
    step's own stream stays un-joined (``join`` still has to be called before IT consumes aux results)."""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    if _dirty.get(idx):
        for st in _streams[idx]:
            stream.wait_stream(st)
