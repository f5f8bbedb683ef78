"""Fluxoid polygons and fluxoid-state solutions (``fluxoid.py:13-119`` of the reference).

Both are callers of the accelerated path: ``find_fluxoid_solution`` is two ``solve(model=...)``
calls around ``Device.mutual_inductance_matrix`` (one warm solve per hole).  The reference builds
the default polygons with shapely (``Polygon.buffer``); shapely is not a dependency here, so
``make_fluxoid_polygons`` offsets the hole outline itself (mitre joins, like the reference's
default ``join_style``) -- exact for the convex, finely sampled holes of the devices this package
targets; pass explicit polygons for anything else.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional, Union

import numpy as np

from .device import Device

logger = logging.getLogger(__name__)


def _segment_distances(p: np.ndarray, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Distance from every point ``p[i]`` to every segment ``a[k] b[k]`` -> ``[len(p), len(a)]``."""
    ab = b - a                                    # [k, 2]
    ap = p[:, None, :] - a[None, :, :]            # [i, k, 2]
    denom = np.maximum(np.einsum("kd,kd->k", ab, ab), 1e-300)
    t = np.clip(np.einsum("ikd,kd->ik", ap, ab) / denom, 0.0, 1.0)
    closest = a[None, :, :] + t[:, :, None] * ab[None, :, :]
    return np.linalg.norm(p[:, None, :] - closest, axis=2)


def _outline_distance(p1: np.ndarray, p2: np.ndarray) -> float:
    """Distance between two closed polygonal outlines that do not cross (the minimum is then
    attained at a vertex of one of them)."""
    a1, b1 = p1[:-1], p1[1:]
    a2, b2 = p2[:-1], p2[1:]
    return float(min(_segment_distances(p1, a2, b2).min(), _segment_distances(p2, a1, b1).min()))


def _offset_mitre(points: np.ndarray, delta: float) -> np.ndarray:
    """Outward offset of a closed counter-clockwise polygon by ``delta`` with mitre joins: every
    edge moves along its outward normal, consecutive offset edges meet at their intersection."""
    pts = points[:-1] if np.allclose(points[0], points[-1]) else points
    nxt = np.roll(pts, -1, axis=0)
    edge = nxt - pts
    length = np.linalg.norm(edge, axis=1)
    keep = length > 0
    pts, edge, length = pts[keep], edge[keep], length[keep]
    normal = np.stack([edge[:, 1], -edge[:, 0]], axis=1) / length[:, None]  # outward for CCW
    n_prev = np.roll(normal, 1, axis=0)
    # vertex i sits between edge i-1 and edge i: move it by delta * (n_prev + n) / (1 + n_prev . n)
    cosang = np.einsum("id,id->i", n_prev, normal)
    scale = delta / np.maximum(1.0 + cosang, 1e-12)
    out = pts + (n_prev + normal) * scale[:, None]
    return np.concatenate([out, out[:1]], axis=0)


def make_fluxoid_polygons(device: Device, holes: Optional[Union[List[str], str]] = None,
                          join_style: str = "mitre", interp_points: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Polygons enclosing the given holes, halfway to the nearest other outline of the same layer
    (``fluxoid.py:13-52``)."""
    if join_style != "mitre":
        raise NotImplementedError("Only mitre joins are implemented without shapely.")
    device_polygons = {**device.films, **device.holes}
    if holes is None:
        holes = list(device.holes)
    if isinstance(holes, str):
        holes = [holes]
    polygons = {}
    for name in holes:
        hole = device.holes[name]
        min_dist = min(_outline_distance(hole.points, other.points)
                       for other in device_polygons.values()
                       if other.layer == hole.layer and other.name != name)
        new_points = _offset_mitre(hole.points, min_dist / 2)
        if interp_points:
            seg = np.linalg.norm(np.diff(new_points, axis=0), axis=1)
            s = np.concatenate([[0.0], np.cumsum(seg)])
            t = np.linspace(0.0, s[-1], interp_points)
            new_points = np.stack([np.interp(t, s, new_points[:, 0]), np.interp(t, s, new_points[:, 1])], axis=1)
        polygons[name] = new_points
    return polygons


def find_fluxoid_solution(model, fluxoids: Optional[Dict[str, float]] = None, **solve_kwargs):
    """The circulating currents that realise the given fluxoid state (units of Phi_0 per hole,
    default 0) and the corresponding :class:`Solution` (``fluxoid.py:55-119``): solve without
    circulating currents, get the mutual-inductance matrix, solve ``M I = target - fluxoids``,
    solve again.  ``hole_polygon_mapping`` may be passed to replace the default polygons."""
    from .solver import solve

    device = model.device
    fluxoids = fluxoids or {}
    hole_names = list(device.holes)
    current_units = model.current_units
    inductance_units = f"Phi_0 / {current_units}"
    solve_kwargs = solve_kwargs.copy()
    applied_field = solve_kwargs.pop("applied_field", None)
    polygons = solve_kwargs.pop("hole_polygon_mapping", None)
    target = np.array([fluxoids.get(name, 0) for name in hole_names], dtype=float)

    orig = dict(model.circulating_currents)
    try:
        model.set_circulating_currents({name: 0 for name in hole_names})
        solution_no_circ = solve(model=model, applied_field=applied_field, **solve_kwargs)[-1]
        if not hole_names:
            if np.any(target):
                raise ValueError("Cannot calculate nonzero fluxoid solution for a device with no holes.")
            return solution_no_circ
        if polygons is None:
            polygons = make_fluxoid_polygons(device)
        current = np.array([
            float(sum(solution_no_circ.hole_fluxoid(name, points=polygons[name], units="Phi_0", with_units=False)))
            for name in hole_names
        ])
        M = device.mutual_inductance_matrix(hole_polygon_mapping=polygons, units=inductance_units, **solve_kwargs)
        I_circ = np.linalg.solve(np.asarray(M.magnitude), target - current)
        model.set_circulating_currents(dict(zip(hole_names, I_circ)))
        solution = solve(model=model, applied_field=applied_field, **solve_kwargs)[-1]
    finally:
        model.set_circulating_currents(orig)
    return solution
