"""Image geometry without OpenCV.

The reference's transforms call ``cv2.getRotationMatrix2D`` / ``cv2.warpAffine`` / ``cv2.flip``
(`/root/reference/custom_transforms.py:116-126`, `:212`) and, through the absent helpers,
``cv2.resize``.  ``cv2`` is not installed here, so the same operations are provided by the native
host runtime (``csrc/host_runtime.cpp``, OpenMP) with a vectorised NumPy implementation of the
identical sampler as a fallback (and as the reference the C++ path is tested against).

Conventions follow OpenCV so transform behaviour carries over: the 2×3 matrix passed to
``warp_affine`` maps SOURCE → DESTINATION (it is inverted internally), ``resize`` uses half-pixel
centres, bicubic uses the Keys kernel with a = −0.75, borders are a constant.
"""
from __future__ import annotations

import ctypes
import math
from typing import Sequence, Tuple

import numpy as np

from ..ops import _backend

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2


def get_rotation_matrix_2d(center: Tuple[float, float], angle_deg: float, scale: float) -> np.ndarray:
    """Same matrix as ``cv2.getRotationMatrix2D`` (positive angle = counter-clockwise, y down)."""
    a = scale * math.cos(math.radians(angle_deg))
    b = scale * math.sin(math.radians(angle_deg))
    cx, cy = center
    return np.array([[a, b, (1 - a) * cx - b * cy],
                     [-b, a, b * cx + (1 - a) * cy]], dtype=np.float64)


def invert_affine(m: np.ndarray) -> np.ndarray:
    a = np.vstack([np.asarray(m, dtype=np.float64), [0.0, 0.0, 1.0]])
    return np.linalg.inv(a)[:2]


def _cubic_w(x: np.ndarray) -> np.ndarray:
    a = -0.75
    x = np.abs(x)
    w1 = ((a + 2) * x - (a + 3)) * x * x + 1
    w2 = ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    return np.where(x <= 1, w1, np.where(x < 2, w2, 0.0))


def _sample_numpy(src: np.ndarray, inv: np.ndarray, hd: int, wd: int, interp: int, border: float) -> np.ndarray:
    hs, ws, c = src.shape
    ys, xs = np.mgrid[0:hd, 0:wd].astype(np.float64)
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]

    def fetch(yy, xx):
        ok = (xx >= 0) & (xx < ws) & (yy >= 0) & (yy < hs)
        v = src[np.clip(yy, 0, hs - 1), np.clip(xx, 0, ws - 1)]
        return np.where(ok[..., None], v, np.float32(border))

    if interp == INTER_NEAREST:
        # C lround: half away from zero
        xi = np.where(sx >= 0, np.floor(sx + 0.5), np.ceil(sx - 0.5)).astype(np.int64)
        yi = np.where(sy >= 0, np.floor(sy + 0.5), np.ceil(sy - 0.5)).astype(np.int64)
        return fetch(yi, xi).astype(np.float32)
    fx0, fy0 = np.floor(sx), np.floor(sy)
    ax, ay = (sx - fx0).astype(np.float32), (sy - fy0).astype(np.float32)
    out = np.zeros((hd, wd, c), dtype=np.float32)
    if interp == INTER_LINEAR:
        offs, wx, wy = (0, 1), (1 - ax, ax), (1 - ay, ay)
    else:
        offs = (-1, 0, 1, 2)
        wx = tuple(_cubic_w(ax - t).astype(np.float32) for t in offs)
        wy = tuple(_cubic_w(ay - t).astype(np.float32) for t in offs)
    for iy, oy in enumerate(offs):
        for ix, ox in enumerate(offs):
            w = (wx[ix] * wy[iy])[..., None]
            out += w * fetch((fy0 + oy).astype(np.int64), (fx0 + ox).astype(np.int64))
    return out


def _sample(src: np.ndarray, inv: np.ndarray, hd: int, wd: int, interp: int, border: float,
            force_numpy: bool = False) -> np.ndarray:
    squeeze = src.ndim == 2
    a = np.ascontiguousarray(src[..., None] if squeeze else src, dtype=np.float32)
    lib = None if force_numpy else _backend.host_lib()
    if lib is not None:
        hs, ws, c = a.shape
        out = np.empty((hd, wd, c), dtype=np.float32)
        m = np.ascontiguousarray(inv, dtype=np.float64).reshape(-1)
        fn = lib.dp_warp_affine_f32
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                       ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_float]
        rc = fn(a.ctypes.data, hs, ws, c, out.ctypes.data, hd, wd, m.ctypes.data, int(interp), float(border))
        if rc != 0:
            raise RuntimeError(f"dp_warp_affine_f32 failed ({rc})")
    else:
        out = _sample_numpy(a, np.asarray(inv, dtype=np.float64), hd, wd, interp, border)
    return out[..., 0] if squeeze else out


def warp_affine(src: np.ndarray, m_fwd: np.ndarray, dsize: Tuple[int, int], interp: int = INTER_LINEAR,
                border_value: float = 0.0, force_numpy: bool = False) -> np.ndarray:
    """``cv2.warpAffine(src, M, (w, h), flags=interp, borderValue=...)`` for float32 HxW[xC]."""
    wd, hd = dsize
    return _sample(src, invert_affine(m_fwd), hd, wd, interp, border_value, force_numpy)


def resize(src: np.ndarray, size_hw: Sequence[int], interp: int = INTER_LINEAR,
           force_numpy: bool = False) -> np.ndarray:
    """Resize to ``(h, w)`` with half-pixel centres; border taps replicate the edge."""
    hd, wd = int(size_hw[0]), int(size_hw[1])
    hs, ws = src.shape[:2]
    if (hs, ws) == (hd, wd):
        return np.asarray(src, dtype=np.float32).copy()
    sx, sy = ws / wd, hs / hd
    inv = np.array([[sx, 0.0, 0.5 * sx - 0.5], [0.0, sy, 0.5 * sy - 0.5]], dtype=np.float64)
    # edge replication: pad by 2 pixels with edge values, shift the map accordingly
    pad = ((2, 2), (2, 2)) + (((0, 0),) if src.ndim == 3 else ())
    padded = np.pad(np.asarray(src, dtype=np.float32), pad, mode="edge")
    inv[:, 2] += 2.0
    return _sample(padded, inv, hd, wd, interp, 0.0, force_numpy)


def flip_horizontal(a: np.ndarray) -> np.ndarray:
    """``cv2.flip(a, 1)``."""
    return np.ascontiguousarray(a[:, ::-1])
