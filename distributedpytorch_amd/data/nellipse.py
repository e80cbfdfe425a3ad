"""``dataloaders.nellipse`` — guidance maps built from the four extreme points of an object.

The reference star-imports this module (`/root/reference/custom_transforms.py:7`) and calls
``compute_nellipse(x_range, y_range, points)`` (`:25`) and
``compute_nellipse_gaussianHM_fast(x_range, y_range, points) -> (z1, z2)`` (`:45`); neither is in
the reference tree, so the definitions below are this framework's own (semantics inferred from the
call sites: maps in 0..1 over a ``len(x_range) × len(y_range)`` canvas, later scaled to 0..255).

An *n-ellipse* is the locus of points whose summed distance to n foci is constant.  With the
extreme points as foci, ``D(p) = Σ_i ‖p − f_i‖`` is smallest inside the object and grows outwards;
the map is 1 at the minimum of D and falls linearly to 0 on the smallest n-ellipse that contains
every focus (``D = max_i D(f_i)``).
"""
from __future__ import annotations

import numpy as np

from .helpers import extreme_points, extreme_points_fixed  # noqa: F401  (star-import surface)

__all__ = ["compute_nellipse", "compute_nellipse_gaussianHM_fast", "extreme_points", "extreme_points_fixed",
           "getPointOfIntersection"]


def _sum_dist(x_range, y_range, points):
    # canvas indexed [row = x_range index, col = y_range index]; points are (x=col, y=row)
    rows = np.asarray(x_range, dtype=np.float32)[:, None]
    cols = np.asarray(y_range, dtype=np.float32)[None, :]
    pts = np.asarray(points, dtype=np.float32).reshape(-1, 2)
    d = np.zeros((rows.shape[0], cols.shape[1]), dtype=np.float32)
    for px, py in pts:
        d += np.sqrt((cols - px) ** 2 + (rows - py) ** 2)
    at_foci = np.array([np.sqrt(((pts - p) ** 2).sum(1)).sum() for p in pts], dtype=np.float32)
    return d, at_foci


def compute_nellipse(x_range, y_range, points) -> np.ndarray:
    d, at_foci = _sum_dist(x_range, y_range, points)
    d_min, d_ref = float(d.min()), float(at_foci.max())
    if d_ref - d_min < 1e-6:
        return (d <= d_ref).astype(np.float32)
    return np.clip(1.0 - (d - d_min) / (d_ref - d_min), 0.0, 1.0).astype(np.float32)


def compute_nellipse_gaussianHM_fast(x_range, y_range, points, sigma: float = 10.0):
    """Returns ``(z1, z2)``: the n-ellipse map and a max-of-Gaussians heat-map at the points."""
    z1 = compute_nellipse(x_range, y_range, points)
    rows = np.asarray(x_range, dtype=np.float32)[:, None]
    cols = np.asarray(y_range, dtype=np.float32)[None, :]
    z2 = np.zeros_like(z1)
    for px, py in np.asarray(points, dtype=np.float32).reshape(-1, 2):
        z2 = np.maximum(z2, np.exp(-4 * np.log(2) * ((cols - px) ** 2 + (rows - py) ** 2) / sigma ** 2))
    return z1, z2.astype(np.float32)


def getPointOfIntersection(points):
    """Intersection of the left–right and top–bottom chords (named in a commented block,
    `custom_transforms.py:131`).  Raises ``ValueError`` when the chords are parallel."""
    (x1, y1), (x2, y2), (x3, y3), (x4, y4) = np.asarray(points, dtype=np.float64).reshape(4, 2)
    den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    if abs(den) < 1e-9:
        raise ValueError("parallel chords")
    a, b = x1 * y2 - y1 * x2, x3 * y4 - y3 * x4
    return np.array([(a * (x3 - x4) - (x1 - x2) * b) / den, (a * (y3 - y4) - (y1 - y2) * b) / den])
