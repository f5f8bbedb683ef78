"""Small geometry helpers kept from the reference's ``geometry.py`` because the solver path and
its fixtures use them: ``circle`` (:60), ``box`` (:104), ``close_curve`` (:182),
``path_vectors`` (:12), ``ensure_unique`` (:198)."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def close_curve(points: np.ndarray) -> np.ndarray:
    """Appends the first point if the curve is not closed (``geometry.py:182-195``)."""
    points = np.asarray(points, dtype=float)
    if not np.array_equal(points[0], points[-1]):
        points = np.concatenate([points, points[:1]], axis=0)
    return points


def ensure_unique(points: np.ndarray) -> np.ndarray:
    """Removes duplicate points, keeping the first occurrence (``geometry.py:198-206``)."""
    _, ix = np.unique(points, return_index=True, axis=0)
    return points[np.sort(ix)]


def path_vectors(path: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Edge lengths and unit normals of a path (``geometry.py:12-29``)."""
    dr = np.diff(path, axis=0)
    lengths = np.linalg.norm(dr, axis=1)
    normals = np.column_stack([dr[:, 1], -dr[:, 0]]) / lengths[:, None]
    return lengths, normals


def circle(radius: float, points: int = 100, center=(0.0, 0.0)) -> np.ndarray:
    """Closed counter-clockwise circle polygon."""
    theta = np.linspace(0, 2 * np.pi, points)
    xy = radius * np.column_stack([np.cos(theta), np.sin(theta)]) + np.asarray(center, dtype=float)
    xy[-1] = xy[0]
    return xy


def box(width: float, height: float = None, points: int = 101, center=(0.0, 0.0)) -> np.ndarray:
    """Closed counter-clockwise rectangle with about ``points`` vertices along its edges."""
    height = width if height is None else height
    per = max(2, points // 4)
    x0, y0 = -width / 2, -height / 2
    t = np.linspace(0, 1, per, endpoint=False)
    xy = np.concatenate([
        np.column_stack([x0 + width * t, y0 * np.ones(per)]),
        np.column_stack([(x0 + width) * np.ones(per), y0 + height * t]),
        np.column_stack([x0 + width * (1 - t), (y0 + height) * np.ones(per)]),
        np.column_stack([x0 * np.ones(per), y0 + height * (1 - t)]),
    ]) + np.asarray(center, dtype=float)
    return close_curve(xy)
