"""``dataloaders.skewed_axes_weight_map`` — confidence / weight maps for ``AddConfidenceMap``.

Star-imported by the reference (`/root/reference/custom_transforms.py:6`) and used at `:283`,
`:289`, `:290`; not shipped, so these are this framework's own definitions with the call-site
signatures: ``generate_mvL1L2_image_skewed_axes(mask, extreme_points=, FULL_IMAGE_WEIGHTS=,
d2_THRESH=, tau=) -> (h_map, d1, d2)``, ``generate_mvgauss_image(mask, FULL_IMAGE_WEIGHTS=, tau=)``
and ``normalize_wtMap(h)``.

The two *skewed axes* are the left→right and top→bottom chords of the extreme points (generally
not orthogonal).  ``d1``/``d2`` are the distances of every pixel to those chords, each normalised
by the half-length of the other chord; the map mixes an L1 and an L2 combination of the two.
"""
from __future__ import annotations

import numpy as np

from .helpers import extreme_points, extreme_points_fixed  # noqa: F401

__all__ = ["generate_mvL1L2_image_skewed_axes", "generate_mvgauss_image", "normalize_wtMap", "extreme_points",
           "extreme_points_fixed"]


def normalize_wtMap(h: np.ndarray) -> np.ndarray:
    h = np.asarray(h, dtype=np.float32)
    lo, hi = float(h.min()), float(h.max())
    return (h - lo) / (hi - lo + 1e-10)


def _dist_to_line(xs, ys, p, q):
    p, q = np.asarray(p, dtype=np.float32), np.asarray(q, dtype=np.float32)
    d = q - p
    n = float(np.hypot(d[0], d[1])) + 1e-6
    return np.abs(d[1] * (xs - p[0]) - d[0] * (ys - p[1])) / n, n


def generate_mvL1L2_image_skewed_axes(mask, extreme_points=None, FULL_IMAGE_WEIGHTS=1, d2_THRESH=None, tau=1.0):
    mask = np.asarray(mask) > 0
    h, w = mask.shape[:2]
    if extreme_points is None:
        extreme_points = extreme_points_fixed(mask.astype(np.float32), 0)
    left, right, top, bottom = np.asarray(extreme_points, dtype=np.float32).reshape(4, 2)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    dist_lr, len_lr = _dist_to_line(xs, ys, left, right)
    dist_tb, len_tb = _dist_to_line(xs, ys, top, bottom)
    d1 = dist_lr / (0.5 * len_tb + 1e-6)   # across the horizontal chord, in units of the vertical half-extent
    d2 = dist_tb / (0.5 * len_lr + 1e-6)
    if d2_THRESH is not None:
        d2 = np.minimum(d2, float(d2_THRESH))
    l1 = d1 + d2
    l2 = np.sqrt(d1 ** 2 + d2 ** 2)
    h_map = np.exp(-0.5 * (l1 + l2) / max(float(tau), 1e-6)).astype(np.float32)
    if not FULL_IMAGE_WEIGHTS:
        h_map = h_map * mask
    return h_map, d1.astype(np.float32), d2.astype(np.float32)


def generate_mvgauss_image(mask, FULL_IMAGE_WEIGHTS=1, tau=0.5):
    """Multivariate Gaussian fitted to the mask pixels (mean + covariance scaled by ``1/tau``)."""
    mask = np.asarray(mask) > 0
    h, w = mask.shape[:2]
    ys, xs = np.where(mask)
    if ys.size < 2:
        return np.zeros((h, w), dtype=np.float32)
    mu = np.array([xs.mean(), ys.mean()])
    cov = np.cov(np.stack([xs, ys]).astype(np.float64)) / max(float(tau), 1e-6) + 1e-3 * np.eye(2)
    icov = np.linalg.inv(cov)
    gy, gx = np.mgrid[0:h, 0:w].astype(np.float64)
    dx, dy = gx - mu[0], gy - mu[1]
    m = icov[0, 0] * dx * dx + 2 * icov[0, 1] * dx * dy + icov[1, 1] * dy * dy
    g = np.exp(-0.5 * m).astype(np.float32)
    return g if FULL_IMAGE_WEIGHTS else g * mask
