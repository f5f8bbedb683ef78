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
    share_mesh: bool = False,
):
    """Builds a :class:`superscreen_amd.Device` made of ``len(kinds)`` coaxial films, one per
    layer at ``z0 = i*z_spacing``, every film meshed with the same ``K``-ring disk triangulation.

    ``kinds[i]`` is ``"disk"`` or ``"washer"`` (disk with a concentric hole).  Every film gets its OWN
    :class:`Mesh` object, as in the reference, where ``Device.make_mesh`` meshes film by film
    (``device/device.py:383-470``); ``share_mesh=True`` hands one object to all films (what rounds 1-5 did: one
    geometry upload and one host-side cache entry then serve every film).
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
    if share_mesh:
        mesh = Mesh.from_triangulation(sites, elements)
        device.meshes = {film.name: mesh for film in films}
    else:
        device.meshes = {film.name: Mesh.from_triangulation(sites.copy(), elements.copy()) for film in films}
    return device


def film_geometry(kind: str, K: int, *, film_radius: float = 5.0, center=(0.0, 0.0)) -> dict:
    """One synthetic film ON ITS OWN MESH: the ``K``-ring disk mesh scaled to ``film_radius`` and moved to
    ``center``.  Returns ``sites, elements, dr, film_polygon, hole_polygon`` (``None`` for a disk) and
    ``fluxoid_polygon`` (a circle half way between hole and rim -- for a disk: at the same radius)."""
    if kind not in ("disk", "washer"):
        raise ValueError(f"Unknown film kind {kind!r}.")
    sites, elements, dr = ring_disk_mesh(K, film_radius)
    center = np.asarray(center, dtype=np.float64)
    Kf = film_rings(K)
    Kh = Kf // 3
    return dict(
        kind=kind, K=int(K), dr=dr, center=center,
        sites=sites + center, elements=elements,
        film_polygon=circle_points((Kf + 0.5) * dr, center=center),
        hole_polygon=circle_points((Kh + 0.5) * dr, 201, center=center) if kind == "washer" else None,
        fluxoid_polygon=circle_points((Kh + (Kf - Kh) / 2 + 0.25) * dr, 101, center=center),
    )


def make_device(films: Sequence[dict], layers: Sequence[dict], *, solve_dtype: str = "float64",
                name: Optional[str] = None):
    """A device whose films are meshed SEPARATELY (the reference's general case, ``solver/solve.py:495-515``:
    ``meshes[source_film].sites`` -> ``meshes[film].sites``; its own multi-film test device is two rings of
    different size, ``test/test_solve.py:40-93``): every film gets its own :class:`Mesh` object, vertex count,
    position and radius, and the layers their own ``Lambda`` and ``z0``; several films may share a layer.

    ``layers``: ``dict(name=, z0=, Lambda=)`` each; ``films``: ``dict(name=, kind=, K=, layer=, film_radius=5.0,
    center=(0, 0))`` each (the keyword arguments of :func:`film_geometry` plus ``name`` and ``layer``).  A washer's
    hole is called ``"hole_" + name``.
    """
    from .device import Device, Layer, Polygon
    from .mesh import Mesh

    lays = [Layer(l["name"], Lambda=l["Lambda"], z0=l["z0"]) for l in layers]
    polys, holes, meshes = [], [], {}
    for spec in films:
        geo = film_geometry(spec["kind"], spec["K"], film_radius=spec.get("film_radius", 5.0),
                            center=spec.get("center", (0.0, 0.0)))
        polys.append(Polygon(spec["name"], layer=spec["layer"], points=geo["film_polygon"]))
        if geo["hole_polygon"] is not None:
            holes.append(Polygon("hole_" + spec["name"], layer=spec["layer"], points=geo["hole_polygon"]))
        meshes[spec["name"]] = Mesh.from_triangulation(geo["sites"], geo["elements"])
    device = Device(name or "synthetic_" + "_".join(s["name"] for s in films), layers=lays, films=polys,
                    holes=holes, length_units="um", solve_dtype=solve_dtype)
    device.meshes = meshes
    return device


# The device of tests/golden/rings_mixed.npz (recorded from the reference by oracle/make_golden.py): the shape of
# the reference's ``two_rings`` test device (a big and a little ring in two layers, the lower one with Lambda = 0)
# with the little ring moved off the axis and a third, disjoint film in the lower ring's layer (dz = 0 coupling).
RINGS_MIXED = dict(
    layers=[dict(name="layer0", z0=0.0, Lambda=0.0), dict(name="layer1", z0=1.0, Lambda=0.2)],
    films=[
        dict(name="big_ring", kind="washer", K=13, layer="layer0", film_radius=7.5, center=(0.0, 0.0)),
        dict(name="little_ring", kind="washer", K=9, layer="layer1", film_radius=5.0, center=(0.8, -0.5)),
        dict(name="side_disk", kind="disk", K=7, layer="layer0", film_radius=2.0, center=(12.0, 3.0)),
    ],
)


def lambda_ramp(x, y, Lambda0: float = 0.2, cx: float = 0.8, cy: float = -0.5):
    """The Lambda(x, y) of the upper layer in the combined mixed-mesh fixture (``rings_mixed_extras.npz``): a ramp
    across the little ring (centred at ``(cx, cy)``) with a quadratic term; positive on the whole film."""
    x, y = np.asarray(x), np.asarray(y)
    return Lambda0 * (1.0 + 0.08 * (x - cx) + 0.01 * (y - cy) ** 2)


# trapped vortices of the combined fixture: (x, y, film, nPhi0)
RINGS_MIXED_VORTICES = ((5.5, 1.0, "big_ring", 1), (12.3, 3.2, "side_disk", -1))


def tilted_field(x, y, z, B0: float = 1.0):
    """The applied field of the mixed-mesh fixtures: not uniform, so that a film's position matters."""
    return B0 * (1.0 + 0.04 * np.asarray(x) - 0.03 * np.asarray(y) + 0.1 * np.asarray(z))


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


# The coupled device of tests/golden/strip_ring.npz: a current-carrying strip (terminals) under a ring on its own mesh
# -- a field coil and a pickup loop, the shape of the reference's susceptometer notebooks.
STRIP_RING = dict(nx=24, ny=10, length=10.0, width=4.0, strip_Lambda=0.3, ring_K=9, ring_radius=3.0, ring_center=(1.0, 0.5),
                  ring_z0=0.6, ring_Lambda=0.15)


def strip_ring_geometry(spec: dict = STRIP_RING) -> dict:
    """Meshes and polygons of the strip + ring device (data only: used by the fixture generator, the oracle tests and
    :func:`make_strip_ring_device`)."""
    try:
        from .geometry import box
    except ImportError:   # (loaded as a stand-alone file by oracle/make_golden.py and the oracle tests)
        import importlib.util
        import os

        _spec = importlib.util.spec_from_file_location("_ssa_geometry", os.path.join(os.path.dirname(__file__), "geometry.py"))
        _geo = importlib.util.module_from_spec(_spec)
        _spec.loader.exec_module(_geo)
        box = _geo.box

    nx, ny, length, width = spec["nx"], spec["ny"], spec["length"], spec["width"]
    sites, elements = strip_mesh(nx, ny, length, width)
    eps = 1e-3 * min(length / nx, width / ny)
    dx = length / nx
    ring = film_geometry("washer", spec["ring_K"], film_radius=spec["ring_radius"], center=spec["ring_center"])
    return dict(
        strip=dict(sites=sites, elements=elements, film_polygon=box(length + 2 * eps, width + 2 * eps, points=401),
                   terminals={"source": box(dx, width + 4 * eps, points=41, center=(-length / 2, 0.0)),
                              "drain": box(dx, width + 4 * eps, points=41, center=(length / 2, 0.0))}),
        ring=ring)


def make_strip_ring_device(spec: dict = STRIP_RING, solve_dtype: str = "float64"):
    """A strip with a ``source`` and a ``drain`` terminal in layer ``base`` (z = 0) and a ring (``hole_ring``) on its
    own mesh in layer ``top``."""
    from .device import Device, Layer, Polygon
    from .mesh import Mesh

    geo = strip_ring_geometry(spec)
    strip, ring = geo["strip"], geo["ring"]
    device = Device(
        "strip_ring",
        layers=[Layer("base", Lambda=spec["strip_Lambda"], z0=0.0), Layer("top", Lambda=spec["ring_Lambda"], z0=spec["ring_z0"])],
        films=[Polygon("strip", layer="base", points=strip["film_polygon"]),
               Polygon("ring", layer="top", points=ring["film_polygon"])],
        holes=[Polygon("hole_ring", layer="top", points=ring["hole_polygon"])],
        terminals={"strip": [Polygon(name, layer="base", points=pts) for name, pts in strip["terminals"].items()]},
        length_units="um", solve_dtype=solve_dtype)
    device.meshes = {"strip": Mesh.from_triangulation(strip["sites"], strip["elements"]),
                     "ring": Mesh.from_triangulation(ring["sites"], ring["elements"])}
    return device
