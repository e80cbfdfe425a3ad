"""Debug-mode consistency checks for the data-parallel runtime (the reference has no
race detection; its latent multi-process hazards are replica drift and rank-divergent control flow).

Enabled with ``DPAMD_DEBUG=1`` (off on the hot path: every check costs a small collective and a host
read).  Each check makes EVERY rank raise together, so no rank is left waiting in a collective.

* ``check_bucket_order``   — every rank launched the same gradient buckets in the same order (a rank
  that skipped a bucket would pair its all-reduce with the wrong one on the others: silent garbage).
* ``check_gradients_equal`` — after the all-reduce all ranks hold bit-identical gradient arenas.
* ``check_replicas_equal``  — parameters (and BN running statistics) have not drifted apart.
* ``check_equal_batches``   — SyncBN on the GPU path assumes equal per-rank B·H·W (it skips the count
  exchange); verified here.
* ``check_streams_joined``  — the auxiliary stream (weight gradients) was joined before the optimiser
  read the gradient arena.
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist

ENABLED = os.environ.get("DPAMD_DEBUG", "0") != "0"


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _spread(value: float, device) -> float:
    """max − min of ``value`` over the ranks (0 when they agree)."""
    if _world() == 1:
        return 0.0
    # a HOST value: over gloo it travels as a host tensor whatever device the caller names — gloo's device-tensor path
    # (stage through the stream, reduce, copy back) stalled about once in ten two-rank runs sharing one GPU, both ranks
    # inside this very all-reduce (profiles/r3_two_rank_test_stop_once.md)
    if not (dist.get_backend() == "nccl"):
        device = torch.device("cpu")
    t = torch.tensor([value, -value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0].item() + t[1].item())


def _comm_device(t: torch.Tensor):
    return t.device if (dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")


def check_bucket_order(order: List[int], device) -> None:
    h = 0.0
    for i, b in enumerate(order):
        h += (i + 1) * 1009.0 + (b + 1) * 7.0 * (i + 1)
    if _spread(h, device) != 0.0 or _spread(float(len(order)), device) != 0.0:
        raise RuntimeError(f"DPAMD_DEBUG: gradient buckets were launched in a rank-dependent order ({order})")


def _checksum(t: torch.Tensor) -> float:
    f = t.detach().double()
    return float((f.sum() + 3.0 * f.abs().sum()).item())


def check_gradients_equal(flat_g: torch.Tensor, arena=None) -> None:
    dev = _comm_device(flat_g)
    if _spread(_checksum(flat_g), dev) != 0.0:
        culprit = ""
        if arena is not None:       # every rank walks the same slots, so the collectives stay paired
            names = []
            for sl in arena.slots:
                if _spread(_checksum(flat_g[sl.offset:sl.offset + sl.numel]), dev) != 0.0:
                    names.append(f"{sl.name}(bucket {sl.bucket})")
            culprit = ": " + ", ".join(names[:12]) + (" ..." if len(names) > 12 else "")
        raise RuntimeError("DPAMD_DEBUG: ranks hold different gradients after the all-reduce" + culprit)


def check_replicas_equal(flat_p: torch.Tensor, buffers: Iterable[torch.Tensor] = ()) -> None:
    dev = _comm_device(flat_p)
    if _spread(_checksum(flat_p), dev) != 0.0:
        raise RuntimeError("DPAMD_DEBUG: model replicas have drifted apart (parameter checksums differ)")
    for i, b in enumerate(buffers):
        if b.is_floating_point() and b.numel() and _spread(_checksum(b), dev) != 0.0:
            raise RuntimeError(f"DPAMD_DEBUG: buffer #{i} differs across ranks (unsynchronised BatchNorm statistics?)")


def check_equal_batches(local_count: int, device) -> None:
    if _spread(float(local_count), device) != 0.0:
        raise RuntimeError("DPAMD_DEBUG: ranks fed different B·H·W to SyncBatchNorm; the GPU path assumes equal "
                           "shards (DistributedSampler drops the ragged tail)")


def check_streams_joined(device) -> None:
    from ..ops import streams
    if device.type == "cuda" and streams._dirty.get(device.index if device.index is not None else
                                                     torch.cuda.current_device()):
        raise RuntimeError("DPAMD_DEBUG: the auxiliary stream was not joined before the optimiser step")
