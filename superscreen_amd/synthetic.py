"""Deterministic synthetic meshes and devices for tests and benchmarks.

The reference meshes films with meshpy/Triangle (``device/utils.py:17-136``), which is a
third-party mesher that is out of scope (and absent from this image).  BASELINE.json's
configs are quoted on *synthetic* meshes of a named vertex count, generated here:

concentric-ring disk -- ring ``k = 1..K`` carries ``6k`` points at radius ``k*dr`` with a
per-ring phase offset of ``0.1*k`` rad, plus the centre, ``N(K) = 1 + 3K(K+1)`` vertices,
triangulated with ``scipy.spatial.Delaunay`` (2-D simplices are CCW).  The film is the disk
of radius ``(K_f + 0.5)*dr`` with ``K_f = floor(K/1.1)``, which mimics the 5 % vacuum buffer
of ``Device.make_mesh`` (``device/device.py:385,445``) and keeps every vertex off the polygon
paths.  A "washer" is the same mesh with a concentric hole of radius ``(K_h + 0.5)*dr``,
``K_h = K_f // 3``.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


def num_vertices(K: int) -> int:
    return 1 + 3 * K * (K + 1)


def film_rings(K: int) -> int:
    return int(np.floor(K / 1.1))


def ring_disk_mesh(K: int, film_radius: float = 5.0) -> Tuple[np.ndarray, np.ndarray, float]:
    """Returns ``(sites (n,2) float64, elements (m,3) int64 CCW, dr)``."""
    from scipy.spatial import Delaunay

    Kf = film_rings(K)
    dr = film_radius / (Kf + 0.5)
    pts = [np.zeros((1, 2))]
    for k in range(1, K + 1):
        theta = 0.1 * k + 2 * np.pi * np.arange(6 * k) / (6 * k)
        pts.append(k * dr * np.column_stack([np.cos(theta), np.sin(theta)]))
    sites = np.concatenate(pts)
    tri = Delaunay(sites)
    elements = tri.simplices.astype(np.int64)
    # Qhull returns CCW simplices in 2-D; make that explicit (and robust).
    p = sites[elements]
    area2 = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (
        p[:, 2, 0] - p[:, 0, 0]
    ) * (p[:, 1, 1] - p[:, 0, 1])
    flip = area2 < 0
    elements[flip] = elements[flip][:, [0, 2, 1]]
    # Drop degenerate slivers on the convex hull, if any.
    elements = elements[np.abs(area2) > 1e-12 * dr * dr]
    if len(np.unique(elements)) != len(sites):
        raise RuntimeError("Delaunay triangulation dropped vertices.")
    return sites, elements, dr


def circle_points(radius: float, num: int = 401, center=(0.0, 0.0)) -> np.ndarray:
    """Closed CCW circle polygon (``geometry.circle`` analogue, ``geometry.py``)."""
    theta = np.linspace(0, 2 * np.pi, num)
    xy = radius * np.column_stack([np.cos(theta), np.sin(theta)]) + np.asarray(center)
    xy[-1] = xy[0]
    return xy


def make_stack_device(
    K: int,
    kinds: Sequence[str] = ("disk",),
    *,
    z_spacing: float = 0.5,
    Lambda: float = 0.1,
    film_radius: float = 5.0,
    solve_dtype: str = "float64",
    name: Optional[str] = None,
):
    """Builds a :class:`superscreen_amd.Device` made of ``len(kinds)`` coaxial films, one per
    layer at ``z0 = i*z_spacing``, every film meshed with the same ``K``-ring disk mesh.

    ``kinds[i]`` is ``"disk"`` or ``"washer"`` (disk with a concentric hole).
    """
    from .device import Device, Layer, Polygon
    from .mesh import Mesh

    sites, elements, dr = ring_disk_mesh(K, film_radius)
    Kf = film_rings(K)
    layers, films, holes = [], [], []
    for i, kind in enumerate(kinds):
        lname = f"layer{i}"
        layers.append(Layer(lname, Lambda=Lambda, z0=i * z_spacing))
        fname = f"{kind}{i}"
        films.append(Polygon(fname, layer=lname, points=circle_points((Kf + 0.5) * dr)))
        if kind == "washer":
            Kh = Kf // 3
            holes.append(
                Polygon(f"hole{i}", layer=lname, points=circle_points((Kh + 0.5) * dr, 201))
            )
        elif kind != "disk":
            raise ValueError(f"Unknown film kind {kind!r}.")
    device = Device(
        name or f"stack_K{K}_" + "_".join(kinds),
        layers=layers,
        films=films,
        holes=holes,
        length_units="um",
        solve_dtype=solve_dtype,
    )
    mesh = Mesh.from_triangulation(sites, elements)  # one mesh object shared by all films
    device.meshes = {film.name: mesh for film in films}
    return device


def strip_mesh(nx: int, ny: int, length: float = 10.0, width: float = 4.0) -> Tuple[np.ndarray, np.ndarray]:
    """Structured ``(nx+1) x (ny+1)`` grid on ``[-length/2, length/2] x [-width/2, width/2]``, every
    cell split into two counter-clockwise triangles (alternating diagonals).  All vertices lie
    inside or on the film outline of :func:`make_strip_device`."""
    xs = np.linspace(-length / 2, length / 2, nx + 1)
    ys = np.linspace(-width / 2, width / 2, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    sites = np.column_stack([X.ravel(), Y.ravel()])

    def vid(i, j):
        return i * (ny + 1) + j

    tris = []
    for i in range(nx):
        for j in range(ny):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
            if (i + j) % 2 == 0:
                tris += [(a, b, c), (a, c, d)]
            else:
                tris += [(a, b, d), (b, c, d)]
    return sites, np.asarray(tris, dtype=np.int64)


def make_strip_device(nx: int = 40, ny: int = 16, *, length: float = 10.0, width: float = 4.0,
                      Lambda: float = 0.3, hole_radius: float = 0.0, solve_dtype: str = "float64"):
    """A current-carrying strip: one film with a ``source`` terminal on its left edge and a
    ``drain`` terminal on its right edge (transport currents, ``solve_film.py:308-390``), optionally
    with a round hole in the middle."""
    from .device import Device, Layer, Polygon
    from .geometry import box
    from .mesh import Mesh

    sites, elements = strip_mesh(nx, ny, length, width)
    eps = 1e-3 * min(length / nx, width / ny)
    film = Polygon("strip", layer="base", points=box(length + 2 * eps, width + 2 * eps, points=401))
    dx = length / nx
    src = Polygon("source", layer="base", points=box(dx, width + 4 * eps, points=41, center=(-length / 2, 0.0)))
    drn = Polygon("drain", layer="base", points=box(dx, width + 4 * eps, points=41, center=(length / 2, 0.0)))
    holes = []
    if hole_radius > 0:
        holes.append(Polygon("hole", layer="base", points=circle_points(hole_radius, 101)))
    device = Device("strip", layers=[Layer("base", Lambda=Lambda, z0=0.0)], films=[film], holes=holes,
                    terminals={"strip": [src, drn]}, length_units="um", solve_dtype=solve_dtype)
    device.meshes = {"strip": Mesh.from_triangulation(sites, elements)}
    return device
