"""Rendezvous, device binding and rank-0 guards.

Implements the reference's TODO items "# export CUDA_VISIBLE_DEVICES", the explicit rendezvous
port, and "# save if master process" (`/root/reference/train_pascal.py:4`, `:7-8`): one process
per GPU, ``RANK``/``LOCAL_RANK``/``WORLD_SIZE``/``MASTER_ADDR``/``MASTER_PORT`` from the
environment (torchrun contract), backend ``nccl`` (= RCCL on ROCm, over xGMI) on GPU and ``gloo``
on CPU.  Failure handling: collectives carry a timeout, and a non-finite loss or
an exception on any rank aborts the whole job instead of dead-locking the others.
"""
from __future__ import annotations

import datetime
import os
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class DistInfo:
    rank: int = 0
    local_rank: int = 0
    world_size: int = 1
    backend: str = "none"
    device: torch.device = torch.device("cpu")

    @property
    def is_master(self) -> bool:
        return self.rank == 0


def init_distributed(backend: Optional[str] = None, timeout_s: int = 600,
                     device: Optional[str] = None) -> DistInfo:
    """Initialise (or re-use) the default process group from torchrun-style environment."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # The host driver of this platform only supports dmabuf IPC: RCCL's peer mappings and hipIpc*MemHandle
    # need HSA_ENABLE_IPC_MODE_LEGACY=0, and ROCr reads HSA_* ONCE, at hsa_init — i.e. at the first HIP call.
    # It is therefore set here, BEFORE torch.cuda.is_available()/set_device() below create the context (the
    # package __init__ sets it too, for callers that touch the GPU before rendezvous), and a multi-rank GPU job
    # whose runtime is already up with another value fails loudly instead of dying later in
    # `hipIpcGetMemHandle: invalid argument`.
    if device != "cpu":
        already_up = torch.cuda.is_initialized()
        if not already_up:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world > 1 and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            raise RuntimeError("multi-rank GPU job: export HSA_ENABLE_IPC_MODE_LEGACY=0 before the first GPU call "
                               "(this process already initialised the GPU runtime without it)"
                               if already_up else "multi-rank GPU job needs HSA_ENABLE_IPC_MODE_LEGACY=0")
    use_gpu = torch.cuda.is_available() and device != "cpu"
    if use_gpu:
        ngpu = torch.cuda.device_count()
        dev = torch.device("cuda", local_rank % max(ngpu, 1))
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")
    if backend is None:
        # DPAMD_DIST_BACKEND=gloo: run the GPU ranks over gloo (rehearsal of the multi-rank GPU path on
        # a box with fewer GPUs than ranks; RCCL refuses two ranks on one device)
        backend = os.environ.get("DPAMD_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    force = os.environ.get("DPAMD_DDP_FORCE", "0") != "0"   # 1-rank process group (see parallel.ddp)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = dict(backend=backend, rank=rank, world_size=world,
                      timeout=datetime.timedelta(seconds=timeout_s))
        if backend == "nccl":
            kwargs["device_id"] = dev
            # (RCCL's kernels on a HIGH-priority stream were tried: the whole gradient exchange is ~1.5 ms of a 130 ms
            # step over the 7 xGMI links, and high-priority queues stalled ranks that share a device
            # (tests/test_gpu_p2p.py rehearsal) — removed with its knob in round 3.)
        dist.init_process_group(**kwargs)
    return DistInfo(rank, local_rank, world, backend if (world > 1 or force) else "none", dev)


def shutdown() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def is_master() -> bool:
    return get_rank() == 0


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


@contextmanager
def master_first():
    """Rank 0 runs the body first (e.g. dataset preprocessing that WRITES a cache file,
    `/root/reference/pascal.py:189`); the others wait, then run it against the finished cache."""
    if not is_master():
        barrier()
    try:
        yield
    finally:
        if is_master():
            barrier()


def _stage_through_host(t: torch.Tensor, group=None) -> bool:
    """gloo cannot be trusted with GPU tensors here (its worker threads block in stream
    synchronisation while other threads launch; 2 ranks sharing one MI355X stall): the multi-rank
    REHEARSAL configuration (gloo backend, GPU tensors) moves the message through the host on the
    calling thread instead.  RCCL ("nccl") never takes this path."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_(t: torch.Tensor, group=None, async_op: bool = False):
    """SUM all-reduce of ``t`` in place; returns a work handle only for ``async_op`` on a backend that
    runs it on its own stream (RCCL)."""
    if _stage_through_host(t, group):
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast_(t: torch.Tensor, src: int = 0, group=None) -> None:
    if _stage_through_host(t, group):
        host = t.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
        return
    dist.broadcast(t, src=src, group=group)


def broadcast_int(value: int, src: int = 0) -> int:
    """``value`` of rank ``src`` on every rank (a one-element message; the caller is off the hot path: tile
    choices of a new problem, ops._hip_conv)."""
    if get_world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_default_device())
    dist.broadcast(t, src=src)
    return int(t.item())


def all_reduce_scalar(value: float, op: str = "sum", device: Optional[torch.device] = None) -> float:
    if get_world_size() == 1:
        return float(value)
    if device is None or dist.get_backend() != "nccl":
        device = _default_device()          # (gloo: a host tensor whatever the caller names, see _stage_through_host)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
    return float(t.item())


def _default_device() -> torch.device:
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def check_finite_or_abort(loss: torch.Tensor, step: int) -> None:
    """Failure detection: every rank learns whether ANY rank saw a non-finite loss, and all of
    them raise together (no rank is left waiting in a collective)."""
    bad = 0.0 if bool(torch.isfinite(loss.detach()).all()) else 1.0
    if all_reduce_scalar(bad, "max", loss.device if loss.is_cuda else None) > 0:
        raise FloatingPointError(f"non-finite loss at step {step} on at least one rank")
