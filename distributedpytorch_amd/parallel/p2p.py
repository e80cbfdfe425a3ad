"""One-shot peer-to-peer all-reduce for small fp32 messages (csrc/p2p_allreduce.hip).

SyncBatchNorm exchanges ``2C`` floats per layer and direction: ~226 latency-bound
collectives per DeepLabv3+ R101 step.  Instead of RCCL's general machinery, every rank maps the
fine-grained receive buffer of every other rank through HIP IPC once, and a collective is ONE small
kernel per rank that stores its vector straight into all peers' buffers over xGMI and sums what
arrives (tagged 8-byte granules, no fences, rank-ordered summation → bit-identical on all ranks).

``DPAMD_SYNCBN_P2P``: ``auto`` (default) | ``1`` | ``0``.  ``auto`` uses the kernel for the default process
group when it is an RCCL group of 2..8 ranks on ONE host and the set-up PROVES itself: every rank maps every
peer's buffer, one collective on a known vector is checked against its closed form, and the ranks agree
(over RCCL) that all of them passed — otherwise every rank falls back to RCCL all-reduces together.  The
set-up handshake (IPC handle exchange) goes through the existing ``torch.distributed`` group.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from ..ops import _backend as B

_ENV = os.environ.get("DPAMD_SYNCBN_P2P", "auto")
# True / False: forced; None: decided once by ``resolve()`` (auto)
ENABLED = None if _ENV == "auto" else (_ENV != "0")
_MAXN = 8192            # floats per message: 2 x 4096 channels
# every spin is bounded: ranks may be skewed by seconds while they autotune in the first step
_TIMEOUT_MS = int(os.environ.get("DPAMD_P2P_TIMEOUT_MS", "30000"))
CHECK_EVERY = 20        # steps between host reads of the error word
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
        self._mine, self._opened = None, []
        # Every stage ends in an all-gather of (payload, error): a rank whose stage failed reports it there, so all
        # ranks leave the constructor together — by raising together — and none waits in a collective for a peer
        # that already gave up.
        mine = ctypes.c_void_p()
        handle = (ctypes.c_ubyte * 64)()
        err = ""
        st = lib.dp_p2p_alloc(nbytes, ctypes.byref(mine))
        if st != 0:
            err = f"p2p_alloc failed with status {st}"
        else:
            self._mine = mine
            st = lib.dp_p2p_get_handle(mine, handle)
            if st != 0:
                err = f"hipIpcGetMemHandle failed with status {st} (HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)"
        # handshake over the existing backend: every rank learns every handle
        gathered = [None] * self.world
        import socket
        where = (socket.gethostname(), os.environ.get("HIP_VISIBLE_DEVICES", ""),
                 os.environ.get("ROCR_VISIBLE_DEVICES", ""), torch.cuda.current_device())
        dist.all_gather_object(gathered, (bytes(handle), os.getpid(), err, where), group=group)
        self._raise_together([g[2] for g in gathered], "set-up")
        self.shared_device = len({g[3] for g in gathered}) < self.world
        if self.shared_device:
            # ranks sharing a GPU (a rehearsal): every stream at one priority, see ops.streams.set_priority
            from ..ops import streams
            streams.set_priority("normal")
        ptrs = (ctypes.c_void_p * self.world)()
        err = ""
        for r, (h, _pid, _e, _w) in enumerate(gathered):
            if r == self.rank:
                ptrs[r] = mine.value
                continue
            hb = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            p = ctypes.c_void_p()
            st = lib.dp_p2p_open_handle(hb, ctypes.byref(p))
            if st != 0 or not p.value:
                err = f"hipIpcOpenMemHandle(rank {r}) failed with status {st}"
                break
            ptrs[r] = p.value
            self._opened.append(p)
        self._ptrs = ptrs
        self._seq = 0
        self._err = torch.zeros(1, dtype=torch.int32, device="cuda")
        # also the barrier: nobody stores into a buffer before everyone has mapped it
        errs = [None] * self.world
        dist.all_gather_object(errs, err, group=group)
        self._raise_together(errs, "peer mapping")

    def _raise_together(self, errs, stage: str) -> None:
        bad = [f"rank {r}: {e}" for r, e in enumerate(errs) if e]
        if bad:
            self.close(sync=False)
            raise RuntimeError(f"one-shot all-reduce {stage} failed on " + "; ".join(bad))

    def all_reduce_(self, t: torch.Tensor, timeout_ms: Optional[int] = None) -> torch.Tensor:
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
                                        _TIMEOUT_MS if timeout_ms is None else int(timeout_ms), B.stream_ptr()),
                "p2p_allreduce")
        return t

    def check(self) -> None:
        """Host read of the error word (call at a synchronisation point, e.g. once per step)."""
        e = int(self._err.item())
        if e:
            raise RuntimeError(f"one-shot all-reduce timed out waiting for a peer (call #{e})")

    def close(self, sync: bool = True) -> None:
        lib = _lib()
        if sync:
            torch.cuda.synchronize()
        for p in self._opened:
            lib.dp_p2p_close_handle(p)
        self._opened = []
        if self._mine is not None:
            lib.dp_p2p_free(self._mine)
            self._mine = None


_default: Optional[OneShotAllReduce] = None
_first_checked = False


def default_instance() -> OneShotAllReduce:
    """Lazily built instance over the default process group (first SyncBN layer of the first step)."""
    global _default
    if _default is None:
        _default = OneShotAllReduce(None)
    return _default


def _same_host(group=None) -> bool:
    import socket
    names = [None] * dist.get_world_size(group)
    dist.all_gather_object(names, socket.gethostname(), group=group)
    return len(set(names)) == 1


def resolve() -> bool:
    """Decide (once, collectively) whether SyncBN statistics travel through the one-shot kernel.  Forced settings
    are returned as they are; ``auto`` = RCCL group of 2..8 ranks on one host + a set-up that passes its self-test
    on EVERY rank.  All ranks call this at the same point (first SyncBN layer of the first forward)."""
    global ENABLED, _default
    if ENABLED is not None:
        return ENABLED
    ok = (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
          and 2 <= dist.get_world_size() <= 8 and torch.cuda.is_available())
    if ok:
        ok = _same_host()
    if not ok:
        ENABLED = False
        return False
    rank, world = dist.get_rank(), dist.get_world_size()
    passed = 0.0
    inst = None
    try:
        inst = OneShotAllReduce(None)                    # raises on ALL ranks or on none
        t = (torch.arange(1027, device="cuda", dtype=torch.float32) * 0.5 + 1.0) * (rank + 1)
        inst.all_reduce_(t, timeout_ms=5000)
        want = (torch.arange(1027, device="cuda", dtype=torch.float32) * 0.5 + 1.0) * (world * (world + 1) // 2)
        torch.cuda.synchronize()
        passed = 1.0 if (torch.equal(t, want) and int(inst._err.item()) == 0) else 0.0
    except Exception as e:      # noqa: BLE001 - any set-up failure means "use RCCL"
        if rank == 0:
            import sys
            print(f"[dpamd] one-shot SyncBN all-reduce unavailable ({e}); using RCCL all-reduces", file=sys.stderr)
        inst = None
    flag = torch.tensor([passed], device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # everyone switches, or nobody does
    ENABLED = bool(flag.item() > 0.5)
    if ENABLED:
        _default = inst
    elif inst is not None:
        inst.close()
        if rank == 0:
            import sys
            print("[dpamd] one-shot SyncBN all-reduce failed its self-test on some rank; using RCCL all-reduces",
                  file=sys.stderr)
    return ENABLED


def check_first_step() -> None:
    """Read the error word at the FIRST step's synchronisation point (then every CHECK_EVERY steps): a reduction
    that timed out returns partial sums, and those must never feed a second step."""
    global _first_checked
    if _default is not None and not _first_checked:
        _first_checked = True
        _default.check()
