"""Loader for the in-tree HIP kernel library (``csrc/libdpamd_hip.so``) and backend policy.

Policy
------
* CUDA(=HIP) tensors run the hand-written gfx950 kernels.  If the shared object is missing on a
  GPU box the first kernel call raises — there is no silent eager fallback.
* CPU tensors (the gloo plumbing configuration, ``BASELINE.json`` config 1) run a plain PyTorch
  implementation of the same math; that implementation is also the fp32 reference the numerics
  tests compare the kernels against.
* ``set_backend('torch')`` forces the PyTorch implementation everywhere.  It exists for exactly two
  things: measuring the stock MIOpen baseline (``bench.py --impl stock``) and
  reference values in tests.
"""
from __future__ import annotations

import ctypes
import os
import sys
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libdpamd_hip.so")
HOST_LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libdpamd_host.so")

_lock = threading.Lock()
_lib = None
_host_lib = None
_mode = os.environ.get("DPAMD_BACKEND", "auto")  # 'auto' | 'torch'


class NativeLibraryMissing(RuntimeError):
    pass


# ---- deterministic mode -----------------------------------------------------------------------------------
# Every cross-block reduction of the kernels (BN statistics in the conv epilogues, fused BN-backward sums,
# split-K weight gradients, depthwise weight gradients, attention dQ / gram, loss sums) is a float atomic: the
# sum depends on the arrival order and two launches of one binary differ in the last bits — bits that a
# BatchNorm+ReLU stack amplifies (profiles/r3_nondeterminism_*.md).  With DPAMD_DETERMINISTIC=1 /
# ``set_deterministic(True)`` no address has two concurrent writers: statistics and depthwise weight gradients get
# one replica per block (summed afterwards by a fixed-order reduction), weight gradients run with split-K = 1,
# the standalone reduction kernels with one row split, attention dQ through per-key-block partials, tile
# configurations come from the packaged table or the default (no timing-dependent autotuning), and the
# auxiliary weight-gradient stream is off.  Two launches are then BIT-identical.  Slower (x1.2-3): a mode for
# bisecting regressions and for tests, not for training runs.
_deterministic = os.environ.get("DPAMD_DETERMINISTIC", "0") not in ("0", "")
_det_synced = False


def deterministic() -> bool:
    return _deterministic


def set_deterministic(flag: bool) -> bool:
    """Turn the deterministic mode on / off; returns the previous setting."""
    global _deterministic, _det_synced
    old, _deterministic = _deterministic, bool(flag)
    _det_synced = False
    if _lib is not None:
        _sync_det()
    return old


def _sync_det() -> None:
    global _det_synced
    if not _det_synced and _lib is not None:
        _lib.dp_set_deterministic(int(_deterministic))
        _det_synced = True


class deterministic_mode:
    """Context manager: ``with deterministic_mode(): ...``"""

    def __init__(self, flag: bool = True):
        self.flag = flag

    def __enter__(self):
        self.prev = set_deterministic(self.flag)
        return self

    def __exit__(self, *exc):
        set_deterministic(self.prev)
        return False


def set_backend(mode: str) -> None:
    global _mode
    if mode not in ("auto", "torch"):
        raise ValueError("backend must be 'auto' or 'torch'")
    _mode = mode


def get_backend() -> str:
    return _mode


class backend:
    """Context manager: ``with backend('torch'): ...``"""

    def __init__(self, mode: str):
        self.mode = mode

    def __enter__(self):
        self.prev = get_backend()
        set_backend(self.mode)
        return self

    def __exit__(self, *exc):
        set_backend(self.prev)
        return False


def use_hip(t: torch.Tensor) -> bool:
    """True when tensor ``t`` must be processed by the HIP kernels."""
    return t.is_cuda and _mode != "torch"


_build_info: dict = {}


def _ensure_current() -> None:
    """Missing or stale ``libdpamd_hip.so`` (its embedded source hash differs from the ``*.hip`` / ``common.h``
    beside it): rebuild in-tree when ``DPAMD_AUTO_BUILD`` allows (default) and hipcc exists, else refuse.  A
    stale library is never loaded silently — kernels and their Python launchers change together."""
    from ..csrc import build as _b
    want, have = _b.source_hash(), _b.embedded_hash(LIB_PATH)
    _build_info.update(source_hash=want, lib_hash=have, rebuilt=False)
    if have == want:
        return
    why = "not found" if not os.path.isfile(LIB_PATH) else f"stale (built from {have or 'unknown sources'}, tree is {want})"
    if os.environ.get("DPAMD_AUTO_BUILD", "1") == "0" or not os.path.isfile(_b.HIPCC):
        raise NativeLibraryMissing(
            f"{LIB_PATH} {why}: run `python -m distributedpytorch_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950) before using the GPU path")
    import sys
    print(f"[dpamd] {LIB_PATH} {why}: building in-tree (hipcc --offload-arch=gfx950) ...", file=sys.stderr, flush=True)
    _b.build_all(verbose=False)
    _build_info.update(lib_hash=_b.embedded_hash(LIB_PATH), rebuilt=True)
    if _build_info["lib_hash"] != want:
        raise NativeLibraryMissing(f"{LIB_PATH}: rebuild did not produce a library matching the sources")


def build_info() -> dict:
    """{'source_hash', 'lib_hash', 'rebuilt'} of the loaded kernel library (bench reports, logs)."""
    if not _build_info:
        from ..csrc import build as _b
        _build_info.update(source_hash=_b.source_hash(), lib_hash=_b.embedded_hash(LIB_PATH), rebuilt=False)
    return dict(_build_info)


def lib() -> ctypes.CDLL:
    """The device-kernel library.  Builds it when missing/stale, raises loudly when it cannot."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                alt = os.environ.get("DPAMD_HIP_LIB")          # A/B runs against a library built from another tree
                if alt:
                    print(f"[dpamd] DPAMD_HIP_LIB: loading {alt} instead of the in-tree library", file=sys.stderr, flush=True)
                    _build_info.update(source_hash="(DPAMD_HIP_LIB)", lib_hash=os.path.basename(alt), rebuilt=False)
                    _lib = ctypes.CDLL(alt)
                else:
                    _ensure_current()
                    _lib = ctypes.CDLL(LIB_PATH)
                for fn in ("dp_set_drop_word", "dp_sgd_set_lr_ptr"):        # captured steps (parallel/engine.py)
                    if hasattr(_lib, fn):
                        getattr(_lib, fn).argtypes = [ctypes.c_void_p]
                        getattr(_lib, fn).restype = ctypes.c_int
                _sync_det()
    return _lib


def host_lib() -> ctypes.CDLL | None:
    """The host-side C++ runtime (CPU image warps, bucket planner).  Optional on CPU."""
    global _host_lib
    if _host_lib is None and os.path.isfile(HOST_LIB_PATH):
        with _lock:
            if _host_lib is None:
                _host_lib = ctypes.CDLL(HOST_LIB_PATH)
    return _host_lib


def lib_built() -> bool:
    return os.path.isfile(LIB_PATH)


def lib_current() -> bool:
    from ..csrc import build as _b
    return _b.is_current()


def stream_ptr() -> ctypes.c_void_p:
    """Current HIP stream as a raw ``hipStream_t`` for the C launchers."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t: torch.Tensor | None) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"HIP launcher '{what}' failed with status {status}")
