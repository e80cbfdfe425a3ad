"""One-shot peer-to-peer all-reduce for small fp32 messages (csrc/p2p_allreduce.hip).

SyncBatchNorm exchanges ``2C`` floats per layer and direction: ~226 latency-bound
collectives per DeepLabv3+ R101 step.  Instead of RCCL's general machinery, every rank maps the
fine-grained receive buffer of every other rank through HIP IPC once, and a collective is ONE small
kernel per rank that stores its vector straight into all peers' buffers over xGMI and sums what
arrives (tagged 8-byte granules, no fences, rank-ordered summation → bit-identical on all ranks).

Opt-in: ``DPAMD_SYNCBN_P2P=1`` makes ``ops.batchnorm`` use it for the default process group on the GPU.
The set-up handshake (IPC handle exchange) goes through the existing ``torch.distributed`` group.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from ..ops import _backend as B

ENABLED = os.environ.get("DPAMD_SYNCBN_P2P", "0") != "0"
_MAXN = 8192            # floats per message: 2 x 4096 channels
# every spin is bounded: ranks may be skewed by seconds while they autotune in the first step
_TIMEOUT_MS = int(os.environ.get("DPAMD_P2P_TIMEOUT_MS", "30000"))
CHECK_EVERY = int(os.environ.get("DPAMD_P2P_CHECK_EVERY", "20"))   # steps between host reads of the error word
_sigs = False


def _lib():
    global _sigs
    lib = B.lib()
    if not _sigs:
        V, I = ctypes.c_void_p, ctypes.c_int
        lib.dp_p2p_alloc.argtypes = [ctypes.c_long, ctypes.POINTER(V)]
        lib.dp_p2p_free.argtypes = [V]
        lib.dp_p2p_get_handle.argtypes = [V, V]
        lib.dp_p2p_open_handle.argtypes = [V, ctypes.POINTER(V)]
        lib.dp_p2p_close_handle.argtypes = [V]
        lib.dp_p2p_allreduce.argtypes = [V, I, ctypes.POINTER(V), I, I, I, ctypes.c_uint, V, I, V]
        _sigs = True
    return lib


class OneShotAllReduce:
    """``all_reduce_(t)``: in-place SUM over the ranks of ``group`` for contiguous fp32 GPU tensors of at
    most ``maxn`` elements.  All ranks must call it in the same order (as any collective)."""

    def __init__(self, group=None, maxn: int = _MAXN):
        if not dist.is_initialized():
            raise RuntimeError("OneShotAllReduce needs an initialised process group for the handshake")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 16:
            raise ValueError("one-shot all-reduce: at most 16 ranks (one node)")
        self.maxn = int(maxn)
        lib = _lib()
        nbytes = 2 * self.world * self.maxn * 8
        mine = ctypes.c_void_p()
        B.check(lib.dp_p2p_alloc(nbytes, ctypes.byref(mine)), "p2p_alloc")
        self._mine = mine
        handle = (ctypes.c_ubyte * 64)()
        B.check(lib.dp_p2p_get_handle(mine, handle), "p2p_get_handle")
        # handshake over the existing backend: every rank learns every handle (and the owner's pid, so a
        # rank sharing the PROCESS with the owner would not re-open its own allocation)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (bytes(handle), os.getpid()), group=group)
        self._opened = []
        ptrs = (ctypes.c_void_p * self.world)()
        for r, (h, _pid) in enumerate(gathered):
            if r == self.rank:
                ptrs[r] = mine.value
                continue
            hb = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            p = ctypes.c_void_p()
            B.check(lib.dp_p2p_open_handle(hb, ctypes.byref(p)), f"p2p_open_handle(rank {r})")
            ptrs[r] = p.value
            self._opened.append(p)
        self._ptrs = ptrs
        self._seq = 0
        self._err = torch.zeros(1, dtype=torch.int32, device="cuda")
        dist.barrier(group=group)      # nobody stores into a buffer before everyone has mapped it

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("one-shot all-reduce: contiguous fp32 GPU tensor")
        n = t.numel()
        if n == 0:
            return t
        if n > self.maxn:
            raise ValueError(f"one-shot all-reduce: {n} > {self.maxn} elements")
        self._seq += 1
        B.check(_lib().dp_p2p_allreduce(B.ptr(t), n, self._ptrs, self.rank, self.world, self.maxn,
                                        ctypes.c_uint(self._seq & 0xFFFFFFFF or 1), B.ptr(self._err),
                                        _TIMEOUT_MS, B.stream_ptr()), "p2p_allreduce")
        return t

    def check(self) -> None:
        """Host read of the error word (call at a synchronisation point, e.g. once per step)."""
        e = int(self._err.item())
        if e:
            raise RuntimeError(f"one-shot all-reduce timed out waiting for a peer (call #{e})")

    def close(self) -> None:
        lib = _lib()
        torch.cuda.synchronize()
        for p in self._opened:
            lib.dp_p2p_close_handle(p)
        self._opened = []
        if self._mine is not None:
            lib.dp_p2p_free(self._mine)
            self._mine = None


_default: Optional[OneShotAllReduce] = None


def default_instance() -> OneShotAllReduce:
    """Lazily built instance over the default process group (first SyncBN layer of the first step)."""
    global _default
    if _default is None:
        _default = OneShotAllReduce(None)
    return _default
