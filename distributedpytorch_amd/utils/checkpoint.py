"""Checkpointing compatible with the reference's key space.

The reference saves ``torch.save(net.state_dict(), path)`` of a ``DataParallel``-wrapped net
(`/root/reference/train_pascal.py:92`, `:230`, `:304`) and loads with
``net.load_state_dict(torch.load(path))`` after wrapping (`:103`) — i.e. a flat
``OrderedDict[str, Tensor]`` whose keys all start with ``module.``, fp32, CPU-loadable.  The weights
file written here is exactly that (contiguous NCHW tensors, regardless of the channels-last arena
layout used for training).  Everything the reference does NOT save — optimiser momentum, epoch,
LR position, RNG, best metric — goes to a *sidecar* ``<path>.state`` so the weights file stays
loadable by the reference.  Rank 0 writes, atomically (tmp + rename), between two barriers (C5).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from ..parallel import dist_env

PREFIX = "module."


def _unwrap(model: nn.Module) -> nn.Module:
    return model.module if hasattr(model, "module") and isinstance(model.module, nn.Module) else model


def reference_state_dict(model: nn.Module) -> "OrderedDict[str, torch.Tensor]":
    """``module.``-prefixed, fp32/int64, contiguous, CPU tensors."""
    from .hostcopy import to_host
    items = list(_unwrap(model).state_dict().items())
    host = to_host([v.detach().float() if v.is_floating_point() else v.detach() for _, v in items])
    return OrderedDict((PREFIX + k, t) for (k, _), t in zip(items, host))


def strip_prefix(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k[len(PREFIX):] if k.startswith(PREFIX) else k, v) for k, v in sd.items())


def _atomic_save(obj: Any, path: str) -> None:
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def rank_state_path(path: str, rank: Optional[int] = None) -> str:
    return f"{path}.state.rank{dist_env.get_rank() if rank is None else rank}"


def save_checkpoint(path: str, model: nn.Module, optimizer=None, extra: Optional[Dict] = None,
                    sync: bool = True, rank_state: Optional[Dict] = None) -> None:
    """Weights → ``path`` (reference-loadable); optimiser/epoch/rank 0's RNG → ``path + '.state'`` (rank 0);
    EVERY rank's own random state → ``path + '.state.rank<r>'``: its torch / NumPy / dropout-counter state plus
    ``rank_state`` (whatever else decides that rank's data stream: the device batch loader's generator, the
    sampler epoch).  A resumed rank takes ITS file back — installing rank 0's state everywhere would give all
    ranks one augmentation stream."""
    if sync:
        dist_env.barrier()
    if dist_env.is_master():
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        _atomic_save(reference_state_dict(model), path)
        side: Dict[str, Any] = {"extra": dict(extra or {}), "world_size": dist_env.get_world_size(),
                                **capture_rng_state()}
        if optimizer is not None:
            side["optimizer"] = optimizer.state_dict()
        _atomic_save(side, path + ".state")
    if sync:
        dist_env.barrier()            # the directory exists before the other ranks write beside it
    if dist_env.get_world_size() > 1 or rank_state is not None:
        _atomic_save({"rank": dist_env.get_rank(), "world_size": dist_env.get_world_size(),
                      "rank_state": dict(rank_state or {}), **capture_rng_state()}, rank_state_path(path))
    if sync:
        dist_env.barrier()


def load_weights(path: str, model: nn.Module, strict: bool = True, map_location="cpu"):
    """Accepts prefixed or un-prefixed keys, a bare state_dict or ``{'state_dict': ...}``
    (`/root/reference/train_pascal.py:103-104`).  Files are read with ``weights_only=True``."""
    obj = torch.load(path, map_location=map_location, weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj and not torch.is_tensor(obj["state_dict"]):
        obj = obj["state_dict"]
    target = _unwrap(model)
    result = target.load_state_dict(strip_prefix(obj), strict=strict)
    arena = getattr(model, "arena", None)
    if arena is not None:
        arena.refresh_lowp()
    return result


def load_checkpoint(path: str, model: nn.Module, optimizer=None, strict: bool = True) -> Dict:
    """Returns the sidecar's ``extra`` dict; ``extra['rank_state']`` is this rank's own ``rank_state`` of the save
    (or ``{}``).  Random state: this rank's file when there is one written by a job of the same size; otherwise
    rank 0 takes the sidecar's and the other ranks KEEP the streams they derived from seed and rank."""
    load_weights(path, model, strict)
    extra: Dict = {}
    side_path = path + ".state"
    world, rank = dist_env.get_world_size(), dist_env.get_rank()
    side = None
    if os.path.isfile(side_path):
        side = torch.load(side_path, map_location="cpu", weights_only=True)
        extra = dict(side.get("extra", {}))
        if optimizer is not None and "optimizer" in side:
            optimizer.load_state_dict(side["optimizer"])
    mine = None
    rp = rank_state_path(path)
    if os.path.isfile(rp):
        cand = torch.load(rp, map_location="cpu", weights_only=True)
        if int(cand.get("world_size", -1)) == world and int(cand.get("rank", -1)) == rank:
            mine = cand
    if mine is not None:
        restore_rng_state(mine)
        extra["rank_state"] = dict(mine.get("rank_state", {}))
    else:
        if side is not None and rank == 0 and int(side.get("world_size", world)) == world:
            restore_rng_state(side)
        extra["rank_state"] = {}
    return extra


def capture_rng_state() -> Dict[str, Any]:
    """Everything that decides the random stream of a resumed run: torch CPU and GPU generators, NumPy's global
    generator (the CPU transforms draw from it) and the fused-dropout call counter (``ops.batchnorm`` derives
    each dropout mask's seed from torch's seed, the rank and that counter — no device RNG state is involved).
    Only tensors / ints / strings, so the sidecar stays loadable with ``weights_only=True``."""
    import numpy as np
    from ..ops import batchnorm as _bn
    st: Dict[str, Any] = {"rng_cpu": torch.get_rng_state(), "torch_seed": int(torch.initial_seed()),
                          "dropout_calls": int(_bn._drop_calls)}
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        st["rng_cuda"] = torch.cuda.get_rng_state()
    name, keys, pos, has_gauss, cached = np.random.get_state()
    st["np_rng"] = {"name": str(name), "keys": torch.from_numpy(keys.astype(np.int64)), "pos": int(pos),
                    "has_gauss": int(has_gauss), "cached": float(cached)}
    return st


def restore_rng_state(side: Dict[str, Any]) -> None:
    import numpy as np
    from ..ops import batchnorm as _bn
    if "rng_cpu" in side:
        torch.set_rng_state(side["rng_cpu"])
    if "rng_cuda" in side and torch.cuda.is_available():
        torch.cuda.set_rng_state(side["rng_cuda"])
    if "dropout_calls" in side:
        _bn._drop_calls = int(side["dropout_calls"])
    npst = side.get("np_rng")
    if npst:
        np.random.set_state((npst["name"], npst["keys"].numpy().astype(np.uint32), int(npst["pos"]),
                             int(npst["has_gauss"]), float(npst["cached"])))


def run_dir(save_dir_root: str, resume_epoch: int = 0) -> str:
    """``run_<N>`` auto-increment as in `/root/reference/train_pascal.py:73-82`, made race-free:
    only rank 0 scans/creates, the choice is broadcast."""
    import glob

    import torch.distributed as dist
    choice = [None]
    if dist_env.is_master():
        if resume_epoch == 0:
            ids = []
            for d in glob.glob(os.path.join(save_dir_root, "run_*")):
                tail = d.rsplit("_", 1)[-1]
                if tail.isdigit():
                    ids.append(int(tail))
            run_id = max(ids) + 1 if ids else 0
        else:
            run_id = 0
        choice[0] = os.path.join(save_dir_root, f"run_{run_id}")
        os.makedirs(os.path.join(choice[0], "models"), exist_ok=True)
    if dist_env.get_world_size() > 1:
        dist.broadcast_object_list(choice, src=0)
    return choice[0]
