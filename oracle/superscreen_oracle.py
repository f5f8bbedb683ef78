"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy/scipy restatement of the reference algorithm for the hot path named in
BASELINE.json (``north_star``): mesh operators -> dense kernel matrix Q -> per-film system
``A = Q.w^T - Lambda.Del2`` -> ``lu_factor(-A)`` -> ``solve_film`` -> inter-film Biot-Savart
-> Jacobi iteration -> fluxoid.  Every function cites the reference ``file:line`` it follows
(paths relative to ``/root/reference/superscreen/``, reference v0.13.0).

Rules (task statement, item 3):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import this module -- always as the checker / baseline, never as the thing shipped;
  * the product package ``superscreen_amd`` never imports it and has no CPU fallback.

Parity pinning: the reference holds NO golden vectors for this path (SURVEY.md section 4:
its tests are physics invariants at 5e-2).  This oracle is therefore pinned against outputs
of the *reference itself*, run in the build container under inert import stubs
(``oracle/_ref_stubs.py``), recorded by ``oracle/make_golden.py`` into ``tests/golden/*.npz``
and re-checked by ``tests/test_oracle_golden.py`` (runs anywhere; the reference itself cannot
travel to the GPU box, the fixtures can).

Third-party arithmetic on the path (not under /root/reference, versions unpinned there,
``setup.py:30-45``): ``scipy.linalg.lu_factor/lu_solve`` (LAPACK getrf/getrs); this oracle
calls the same routines (scipy 1.15.3 / numpy 2.2.6 in this image).
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as la
import scipy.sparse as sp

# --------------------------------------------------------------------------------------
# Unit constants (reference: pint registry, units.py:1-3; solver/utils.py:407-437).
# CODATA 2018 values as shipped with pint's default registry.
# --------------------------------------------------------------------------------------
MU_0 = 1.25663706212e-6  # N / A^2
PHI_0 = 2.067833848461929e-15  # Wb  (h / 2e)


def field_conversion_mT_to_uA_per_um() -> float:
    """``field_conversion_factor("mT", "uA", "um")`` (solver/utils.py:407-437):
    B = 1 mT -> H = B/mu0 in A/m; 1 A/m == 1 uA/um."""
    return 1e-3 / MU_0


def vortex_flux_uA_um() -> float:
    """``ureg("Phi_0 / mu_0").to("uA * um")`` (solver/solve.py:441-442)."""
    return PHI_0 / MU_0 * 1e6 * 1e6


# --------------------------------------------------------------------------------------
# Mesh geometry (device/utils.py, device/mesh.py)
# --------------------------------------------------------------------------------------
def triangle_areas(points: np.ndarray, triangles: np.ndarray) -> np.ndarray:
    """device/utils.py:230-248 -- signed area 1/2 det[[p2-p1],[p0-p2]]."""
    xy = points[triangles]
    s = xy[:, [2, 0]] - xy[:, [1, 2]]
    return 0.5 * (s[:, 0, 0] * s[:, 1, 1] - s[:, 0, 1] * s[:, 1, 0])


def vertex_areas(points: np.ndarray, triangles: np.ndarray,
                 tri_areas: Optional[np.ndarray] = None) -> np.ndarray:
    """device/utils.py:251-273 -- w_i = 1/3 sum of adjacent triangle areas."""
    if tri_areas is None:
        tri_areas = triangle_areas(points, triangles)
    third = tri_areas / 3
    n = len(points)
    out = np.zeros(n, dtype=float)
    for c in range(3):
        out += np.bincount(triangles[:, c], weights=third, minlength=n)
    return out


def get_edges(triangles: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """device/utils.py:139-152."""
    edges = np.concatenate([triangles[:, e] for e in [(0, 1), (1, 2), (2, 0)]])
    edges = np.sort(edges, axis=1)
    edges, counts = np.unique(edges, return_counts=True, axis=0)
    return edges, counts == 1


def find_boundary_indices(triangles: np.ndarray) -> np.ndarray:
    """device/mesh.py:158-170 -- vertices on edges that belong to one triangle only."""
    edges, is_boundary = get_edges(triangles)
    return np.unique(edges[is_boundary].ravel())


# --------------------------------------------------------------------------------------
# FEM operators (fem.py)
# --------------------------------------------------------------------------------------
def _corner_angles(points: np.ndarray, triangles: np.ndarray, corner: int) -> np.ndarray:
    """Interior angle of every triangle at its ``corner``-th vertex, computed exactly the
    way fem.py:188-224 does (arccos of the normalised dot product)."""
    a, b, c = corner, (corner + 1) % 3, (corner + 2) % 3
    v1 = points[triangles[:, b]] - points[triangles[:, a]]
    v2 = points[triangles[:, c]] - points[triangles[:, a]]
    cosang = np.sum(v1 * v2, axis=1) / (la.norm(v1, axis=1) * la.norm(v2, axis=1))
    return np.arccos(cosang)


def weights_half_cotangent(points: np.ndarray, triangles: np.ndarray) -> sp.csr_array:
    """fem.py:165-224 -- W_ij += 1/2 cot(angle opposite edge ij), symmetric."""
    n = len(points)
    rows, cols, vals = [], [], []
    for corner in range(3):
        w = 0.5 / np.tan(_corner_angles(points, triangles, corner))
        i = triangles[:, (corner + 1) % 3]
        j = triangles[:, (corner + 2) % 3]
        rows += [i, j]
        cols += [j, i]
        vals += [w, w]
    W = sp.coo_array(
        (np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)
    )
    return W.tocsr()


def laplace_operator(points: np.ndarray, triangles: np.ndarray,
                     masses: Optional[np.ndarray] = None) -> sp.csr_array:
    """fem.py:259-296 -- L = W - diag(rowsum W);  Del2 = diag(1/masses) @ L (row scaling)."""
    if masses is None:
        masses = vertex_areas(points, triangles)
    W = weights_half_cotangent(points, triangles).tolil()
    W.setdiag(0)
    W = W.tocsr()
    W.eliminate_zeros()
    rowsum = np.asarray(W.sum(axis=1)).ravel()
    L = (W - sp.diags_array(rowsum, format="csr")).tocsr()
    return (sp.diags_array(1.0 / masses, format="csr") @ L).tocsr()


def gradient_triangles(points: np.ndarray, triangles: np.ndarray,
                       areas: Optional[np.ndarray] = None) -> Tuple[sp.csr_array, sp.csr_array]:
    """fem.py:299-347 -- per-triangle linear gradient, 3 nnz/row."""
    if areas is None:
        areas = triangle_areas(points, triangles)
    xy = points[triangles]
    edges = np.roll(xy, 2, axis=1) - np.roll(xy, 1, axis=1)
    rot = np.empty_like(edges)
    rot[:, :, 0] = +edges[:, :, 1]
    rot[:, :, 1] = -edges[:, :, 0]
    data = (rot / (2 * areas[:, None, None])).reshape(-1, 2).T
    m, n = len(triangles), len(points)
    row = np.repeat(np.arange(m), 3)
    col = triangles.ravel()
    Gx = sp.csr_array((data[0], (row, col)), shape=(m, n), dtype=float)
    Gy = sp.csr_array((data[1], (row, col)), shape=(m, n), dtype=float)
    return Gx, Gy


def gradient_vertices(points: np.ndarray, triangles: np.ndarray,
                      gradient_tri: Optional[Tuple[sp.csr_array, sp.csr_array]] = None
                      ) -> Tuple[sp.csr_array, sp.csr_array]:
    """fem.py:350-402 -- vertex gradient = weighted mean of adjacent triangle gradients.

    Quirk reproduced (fem.py:393-399): the weight of triangle t for vertex i is the angle at
    t's *first* vertex (``vec1 = p[t1]-p[t0]``, ``vec2 = p[t2]-p[t0]``), whatever i is,
    normalised over the triangles adjacent to i.
    """
    if gradient_tri is None:
        gradient_tri = gradient_triangles(points, triangles)
    Gx, Gy = gradient_tri
    m, n = len(triangles), len(points)
    angle0 = _corner_angles(points, triangles, 0)
    vert = triangles.ravel()
    tri = np.repeat(np.arange(m), 3)
    wsum = np.bincount(vert, weights=angle0[tri], minlength=n)
    W = sp.csr_array((angle0[tri] / wsum[vert], (vert, tri)), shape=(n, m))
    return (W @ Gx).tocsr(), (W @ Gy).tocsr()


# --------------------------------------------------------------------------------------
# Kernel matrix (distance.py, device/mesh.py)
# --------------------------------------------------------------------------------------
def q_matrix(points: np.ndarray, block: int = 2048) -> np.ndarray:
    """distance.py:87-115 -- q_ij = (1/4pi) |r_i - r_j|^-3, q_ii = 0; dtype of ``points``."""
    n = len(points)
    out = np.empty((n, n), dtype=points.dtype)
    one_over_4pi = 1 / (4 * np.pi)
    x, y = points[:, 0], points[:, 1]
    for i0 in range(0, n, block):
        i1 = min(n, i0 + block)
        dx = x[i0:i1, None] - x[None, :]
        dy = y[i0:i1, None] - y[None, :]
        r2 = dx * dx + dy * dy
        with np.errstate(divide="ignore"):
            blk = one_over_4pi * r2 ** (-1.5)
        blk[np.arange(i1 - i0), np.arange(i0, i1)] = 0.0
        out[i0:i1] = blk
    return out


def C_vector(points: np.ndarray) -> np.ndarray:
    """device/mesh.py:401-432 -- edge vector; centred on the *mean* of the coordinates
    (:421-422), ``inf -> 1e30`` before the division by 4 pi (:430-431)."""
    x = points[:, 0] - points[:, 0].mean()
    y = points[:, 1] - points[:, 1].mean()
    a = np.ptp(x) / 2
    b = np.ptp(y) / 2
    with np.errstate(divide="ignore"):
        C = sum(
            np.sqrt((a - p * x) ** (-2) + (b - q * y) ** (-2))
            for p, q in itertools.product((-1, 1), repeat=2)
        )
    C[np.isinf(C)] = 1e30
    C /= 4 * np.pi
    return C


def Q_matrix(points: np.ndarray, weights: np.ndarray) -> np.ndarray:
    """device/mesh.py:435-458 -- Q_ij = -q_ij (i != j); Q_ii = (C_i + sum_l q_il w_l)/w_i."""
    q = q_matrix(points)
    C = C_vector(points)
    diag = -(C + np.einsum("ij, j -> i", q, weights)) / weights
    np.fill_diagonal(q, diag)
    return -q


# --------------------------------------------------------------------------------------
# Mesh operators container (device/mesh.py:326-394)
# --------------------------------------------------------------------------------------
@dataclass
class OracleMesh:
    sites: np.ndarray
    elements: np.ndarray
    boundary_indices: np.ndarray
    triangle_areas: np.ndarray
    weights: np.ndarray  # vertex areas
    Q: Optional[np.ndarray]
    laplacian: sp.csr_array
    gradient_x: sp.csr_array
    gradient_y: sp.csr_array
    gradient_tri_x: sp.csr_array
    gradient_tri_y: sp.csr_array


def make_mesh(sites: np.ndarray, elements: np.ndarray, build_Q: bool = True) -> OracleMesh:
    """``Mesh.from_triangulation`` + ``MeshOperators.from_mesh`` (device/mesh.py:111-155,
    362-394)."""
    sites = np.asarray(sites, dtype=float)
    elements = np.asarray(elements, dtype=np.int64)
    tri_a = triangle_areas(sites, elements)
    w = vertex_areas(sites, elements, tri_a)
    Gx, Gy = gradient_triangles(sites, elements, tri_a)
    gx, gy = gradient_vertices(sites, elements, (Gx, Gy))
    return OracleMesh(
        sites=sites,
        elements=elements,
        boundary_indices=find_boundary_indices(elements),
        triangle_areas=tri_a,
        weights=w,
        Q=Q_matrix(sites, w) if build_Q else None,
        laplacian=laplace_operator(sites, elements, w),
        gradient_x=gx,
        gradient_y=gy,
        gradient_tri_x=Gx,
        gradient_tri_y=Gy,
    )


# --------------------------------------------------------------------------------------
# Per-film linear systems (solver/solve_film.py)
# --------------------------------------------------------------------------------------
def build_system_2d(Q, weights, Lambda, laplacian, ix) -> np.ndarray:
    """solver/solve_film.py:296-305 (homogeneous Lambda):
    ``A = Q[ix,ix]*w[ix] - Lambda[ix]*Del2[ix,ix]`` -- COLUMN scalings w_j, Lambda_j."""
    sub = laplacian[ix][:, ix]
    sub = sub.toarray() if sp.issparse(sub) else np.asarray(sub)
    return Q[np.ix_(ix, ix)] * weights[ix] - Lambda[ix] * sub


def build_system_1d(Q, weights, Lambda, laplacian, ix) -> np.ndarray:
    """solver/solve_film.py:285-293: ``A_h = Q[:,ix]*w[ix] - Lambda[ix]*Del2[:,ix]``."""
    sub = laplacian[:, ix]
    sub = sub.toarray() if sp.issparse(sub) else np.asarray(sub)
    return Q[:, ix] * weights[ix] - Lambda[ix] * sub


@dataclass
class OracleFilm:
    """What ``make_film_info`` + ``factorize_linear_systems`` hold for one film
    (solver/utils.py:234-324, solver/solve_film.py:151-282)."""

    name: str
    mesh: OracleMesh
    z0: float
    Lambda: np.ndarray  # (n,)
    film_indices: np.ndarray  # indices of unknowns (interior minus holes), sorted
    hole_indices: Dict[str, np.ndarray]
    dtype: np.dtype
    weights: np.ndarray = None
    Q: np.ndarray = None
    A: np.ndarray = None
    lu_piv: Tuple[np.ndarray, np.ndarray] = None
    A_holes: Dict[str, np.ndarray] = field(default_factory=dict)
    # films with terminals (solver/solve_film.py:220-263)
    boundary_indices: Optional[np.ndarray] = None            # ordered counter-clockwise
    terminal_masks: Optional[Dict[str, np.ndarray]] = None   # {terminal: contains(boundary points)}
    A_boundary: Optional[np.ndarray] = None
    fwb_indices: Optional[np.ndarray] = None                 # film without boundary (holes included)
    fwb_lu_piv: Optional[Tuple[np.ndarray, np.ndarray]] = None


def make_film(name: str, mesh: OracleMesh, *, z0: float, Lambda, in_film: np.ndarray,
              holes_mask: Optional[Dict[str, np.ndarray]] = None,
              dtype="float64", factorize: bool = True,
              boundary_indices: Optional[np.ndarray] = None,
              terminal_masks: Optional[Dict[str, np.ndarray]] = None) -> OracleFilm:
    """``make_film_info`` index logic (solver/utils.py:271-304) followed by
    ``factorize_linear_systems`` (solver/solve_film.py:209-218, 269-281).

    ``in_film`` / ``holes_mask[name]`` are the boolean results of the film / hole polygons'
    ``contains_points(mesh.sites)``.
    """
    dtype = np.dtype(dtype)
    n = len(mesh.sites)
    Lambda = np.broadcast_to(np.asarray(Lambda, dtype=float), (n,)).astype(dtype)
    if np.any(Lambda < 0):
        raise ValueError(f"Negative Lambda in film {name!r}.")  # solver/utils.py:57-58
    holes_mask = holes_mask or {}
    hole_indices = {h: np.where(m)[0] for h, m in holes_mask.items()}
    has_terminals = terminal_masks is not None
    boundary = mesh.boundary_indices if boundary_indices is None else np.asarray(boundary_indices)
    interior_all = np.setdiff1d(np.where(in_film)[0], boundary)  # solver/utils.py:300-303
    interior = interior_all
    if hole_indices:
        interior = np.setdiff1d(interior, np.concatenate(list(hole_indices.values())))
    weights = mesh.weights.astype(dtype, copy=False)
    Q = mesh.Q.astype(dtype, copy=False)
    lap = mesh.laplacian.astype(dtype)
    film = OracleFilm(name=name, mesh=mesh, z0=float(z0), Lambda=Lambda,
                      film_indices=interior, hole_indices=hole_indices, dtype=dtype,
                      weights=weights, Q=Q)
    # inhomogeneous Lambda: LambdaInfo's criterion (solver/utils.py:46-49) and the extra term
    # grad_Lambda_term = einsum("ijk, ijk -> jk", grad @ Lambda, grad)   (solve_film.py:181-185)
    inhomogeneous = np.ptp(Lambda) / max(np.min(np.abs(Lambda)), np.finfo(float).eps) > 1e-6
    term = None
    if inhomogeneous:
        gx = mesh.gradient_x.toarray().astype(dtype, copy=False)
        gy = mesh.gradient_y.toarray().astype(dtype, copy=False)
        term = (gx @ Lambda)[:, None] * gx + (gy @ Lambda)[:, None] * gy
    for h, ix in hole_indices.items():
        A_h = build_system_1d(Q, weights, Lambda, lap, ix)
        if term is not None:
            A_h = A_h - term[:, ix]  # :289-293
        film.A_holes[h] = A_h.astype(dtype, copy=False)
    A = build_system_2d(Q, weights, Lambda, lap, interior)
    if term is not None:
        A = A - term[np.ix_(interior, interior)]  # :300-305
    film.A = A.astype(dtype, copy=False)
    if factorize:
        film.lu_piv = la.lu_factor(-film.A)
    if has_terminals:  # solve_film.py:220-263 (homogeneous Lambda)
        film.boundary_indices = boundary
        film.terminal_masks = terminal_masks
        film.A_boundary = build_system_1d(Q, weights, Lambda, lap, boundary).astype(dtype, copy=False)
        film.fwb_indices = interior_all
        if hole_indices:
            film.fwb_lu_piv = la.lu_factor(-build_system_2d(Q, weights, Lambda, lap, interior_all))
        else:
            film.fwb_lu_piv = film.lu_piv
    return film


def make_films(layers: Sequence[dict], films: Sequence[dict], geometries: Dict[str, dict],
               dtype="float64", meshes: Optional[Dict[str, OracleMesh]] = None,
               lambda_funcs: Optional[Dict[str, object]] = None) -> List[OracleFilm]:
    """The films of a device in which EVERY film has its own mesh (the general case of ``make_film_info``,
    solver/utils.py:244-246: ``mesh = device.meshes[name]``, ``layer = device.layers[film.layer]``).

    ``layers``: ``dict(name=, z0=, Lambda=)``; ``films``: ``dict(name=, layer=, ...)``; ``geometries[name]``:
    ``dict(sites=, elements=, film_polygon=, hole_polygon= | None)`` (what
    ``superscreen_amd.synthetic.film_geometry`` returns -- data only).  A film's hole is ``"hole_" + name``; hole
    membership is ``Polygon.contains_points`` = matplotlib ``Path.contains_points`` (device/polygon.py:159).
    ``lambda_funcs[layer]``: a ``Lambda(x, y)`` evaluated on the film's own sites (solver/utils.py:263-266)."""
    from matplotlib.path import Path

    layer = {l["name"]: l for l in layers}
    out = []
    for f in films:
        geo = geometries[f["name"]]
        mesh = meshes[f["name"]] if meshes is not None else make_mesh(geo["sites"], geo["elements"])
        holes = {}
        if geo.get("hole_polygon") is not None:
            holes["hole_" + f["name"]] = Path(geo["hole_polygon"], closed=True).contains_points(mesh.sites)
        Lam = layer[f["layer"]]["Lambda"]
        if lambda_funcs and f["layer"] in lambda_funcs:
            Lam = lambda_funcs[f["layer"]](mesh.sites[:, 0], mesh.sites[:, 1])
        out.append(make_film(f["name"], mesh, z0=layer[f["layer"]]["z0"], Lambda=Lam,
                             in_film=Path(geo["film_polygon"], closed=True).contains_points(mesh.sites),
                             holes_mask=holes, dtype=dtype))
    return out


@dataclass
class OracleFilmSolution:
    """``FilmSolution`` (solution.py:95-130)."""

    stream: np.ndarray
    current_density: np.ndarray
    applied_field: np.ndarray
    self_field: np.ndarray
    field_from_other_films: Optional[np.ndarray] = None

    @property
    def total_field(self) -> np.ndarray:
        out = self.applied_field + self.self_field
        if self.field_from_other_films is not None:
            out = out + self.field_from_other_films
        return out


def boundary_vertices(points: np.ndarray, triangles: np.ndarray) -> np.ndarray:
    """device/utils.py:205-226 without shapely: the directed boundary edges of the (CCW) triangles
    chained into the outer loop, from its lowest vertex index (see superscreen_amd.fem)."""
    tri = np.asarray(triangles, dtype=np.int64)
    directed = {}
    twins = set()
    for t in tri:
        p = points[t]
        if (p[1, 0] - p[0, 0]) * (p[2, 1] - p[0, 1]) - (p[2, 0] - p[0, 0]) * (p[1, 1] - p[0, 1]) < 0:
            t = t[[0, 2, 1]]
        for k in range(3):
            twins.add((int(t[(k + 1) % 3]), int(t[k])))
            directed[(int(t[k]), int(t[(k + 1) % 3]))] = True
    nxt = {a: b for (a, b) in directed if (a, b) not in twins}
    start = min(nxt)
    loop = [start]
    while nxt[loop[-1]] != start:
        loop.append(nxt[loop[-1]])
    return np.asarray(loop, dtype=np.int64)


def roll_boundary_outside_terminals(indices: np.ndarray, terminal_contains) -> np.ndarray:
    """device/device.py:491-500: roll the loop so that it does not wrap around inside a terminal.
    ``terminal_contains``: list of callables ``f(boundary_positions_order) -> index array``."""
    for contains_ix in terminal_contains:
        t_ix = contains_ix(indices)
        discont = np.diff(t_ix) != 1
        if np.any(discont):
            return np.roll(indices, -(np.where(discont)[0][0] + 1))
    return indices


def _path_vectors(path: np.ndarray):
    """geometry.py:12-29: edge lengths and unit normals (dy, -dx) / |d| of a path."""
    dr = np.diff(path, axis=0)
    lengths = la.norm(dr, axis=1)
    return lengths, np.column_stack([dr[:, 1], -dr[:, 0]]) / lengths[:, None]


def stream_from_terminal_current(points: np.ndarray, current: float) -> np.ndarray:
    """solver/utils.py:440-488: g = cumulative trapezoid of (z x J) . dl for a current density
    that is uniform along and perpendicular to the terminal, normalised to ``current``."""
    from scipy import integrate

    lengths, normals = _path_vectors(points)
    J = current * normals / np.sum(lengths)
    zxJ = np.column_stack([-J[:, 1], J[:, 0]])
    g = integrate.cumulative_trapezoid(np.sum(zxJ * np.diff(points, axis=0), axis=1), initial=0)
    return g * current / g[-1]


def terminal_current_stream(film: OracleFilm, terminal_currents: Dict[str, float]) -> np.ndarray:
    """solve_for_terminal_current_stream (solver/solve_film.py:308-390)."""
    pts = film.mesh.sites
    n = len(pts)
    if not any(terminal_currents.values()):
        return np.zeros(n)
    b = film.boundary_indices
    g = np.zeros(n)
    for tname, mask in film.terminal_masks.items():
        ixb = np.sort(np.where(mask)[0])
        remaining = b[ixb[-1]:]
        ixt = b[ixb]
        stream = stream_from_terminal_current(pts[ixt], -terminal_currents[tname])
        g[ixt[:-1]] += stream
        g[remaining] += stream[-1]
    g = g - np.max(g) + np.ptp(g) / 2
    Ha = -(film.A_boundary @ g[b])
    g[film.fwb_indices] = la.lu_solve(film.fwb_lu_piv, -Ha[film.fwb_indices])
    if not film.hole_indices:
        return g
    Ha = np.zeros(n)
    for h, ix in film.hole_indices.items():
        g[ix] = np.average(g[ix], weights=film.weights[ix])
        Ha += -(film.A_holes[h] @ g[ix])
    Ha += -(film.A_boundary @ g[b])
    g[film.film_indices] = la.lu_solve(film.lu_piv, -Ha[film.film_indices])
    return g


def boundary_effective_field(sites, centers, lengths, normals, stream) -> np.ndarray:
    """_get_boundary_effective_field (solver/solve_film.py:393-412)."""
    dr = sites[:, None, :] - centers[None, :, :]
    r3 = la.norm(dr, axis=2) ** 3
    return np.sum(stream[None, :] / r3 * np.sum(dr * -normals[None, :, :], axis=2) * lengths[None, :], axis=1) / (4 * np.pi)


def biot_savart_within_film(sites, centroids, areas, J_tri) -> np.ndarray:
    """_biot_savart_within_film (solver/solve_film.py:415-437)."""
    dx = sites[:, 0, None] - centroids[None, :, 0]
    dy = sites[:, 1, None] - centroids[None, :, 1]
    pref = areas[None, :] * (dx * dx + dy * dy) ** (-1.5)
    return (np.sum(pref * J_tri[None, :, 0] * dy, axis=1) - np.sum(pref * J_tri[None, :, 1] * dx, axis=1)) / (4 * np.pi)


def solve_film(film: OracleFilm, applied_field: np.ndarray, *, field_conversion: float,
               circulating_currents: Optional[Dict[str, float]] = None,
               field_from_other_films: Optional[np.ndarray] = None,
               vortices: Sequence[Tuple[float, float, float]] = (),
               vortex_flux: Optional[float] = None,
               terminal_currents: Optional[Dict[str, float]] = None) -> OracleFilmSolution:
    """solver/solve_film.py:440-574 without terminals.  ``vortices``: ``(x, y, nPhi0)`` triples
    located in this film (:541-554)."""
    circulating_currents = circulating_currents or {}
    Hz = applied_field
    if field_from_other_films is not None:
        Hz = Hz + field_from_other_films
    g = np.zeros_like(Hz)
    Ha_eff = np.zeros_like(Hz)
    for hname, ix in film.hole_indices.items():  # :498-503 (also when I_circ == 0)
        g[ix] += circulating_currents.get(hname, 0)
        Ha_eff += -(film.A_holes[hname] @ g[ix])
    has_terminals = film.terminal_masks is not None
    if has_terminals:  # :505-524
        g_tr = terminal_current_stream(film, terminal_currents or {})
        g = g + g_tr
        bs = film.mesh.sites[film.boundary_indices]
        stream = g_tr[film.boundary_indices]
        centers = 0.5 * (bs + np.roll(bs, -1, axis=0))
        stream = 0.5 * (stream + np.roll(stream, -1, axis=0))
        lengths, normals = _path_vectors(np.concatenate([bs, bs[:1]]))
        Ha_eff = Ha_eff + boundary_effective_field(film.mesh.sites, centers, lengths, normals, stream)
    ix = film.film_indices
    h = Hz[ix] - Ha_eff[ix]
    gf = la.lu_solve(film.lu_piv, h)  # :530  => g = -A^-1 h
    g[ix] += gf
    K = None
    for (vx, vy, nPhi0) in vortices:  # :541-554
        if K is None:
            K = -la.lu_solve(film.lu_piv, np.eye(film.A.shape[0]))  # = inv(A)
        pts = film.mesh.sites
        j_film = np.argmin(la.norm(pts[ix] - (vx, vy), axis=1))
        j_device = np.argmin(la.norm(pts - (vx, vy), axis=1))
        vf = vortex_flux_uA_um() if vortex_flux is None else vortex_flux
        g[ix] += vf * nPhi0 * K[:, j_film] / film.weights[j_device]  # Eq. 28 in [Brandt]
    J = np.array([film.mesh.gradient_y @ g, -(film.mesh.gradient_x @ g)]).T  # :556
    if has_terminals:  # :557-562
        m = film.mesh
        J_tri = np.array([m.gradient_tri_y @ g, -(m.gradient_tri_x @ g)]).T
        screening = biot_savart_within_film(m.sites, m.sites[m.elements].sum(axis=1) / 3, m.triangle_areas, J_tri)
    else:
        screening = film.Q @ (film.weights * g)  # :565
    other = None
    if field_from_other_films is not None:
        other = field_from_other_films / field_conversion
    return OracleFilmSolution(
        stream=g,
        current_density=J,
        applied_field=applied_field / field_conversion,
        self_field=screening / field_conversion,
        field_from_other_films=other,
    )


# --------------------------------------------------------------------------------------
# Inter-film coupling and the Jacobi loop (solver/solve.py)
# --------------------------------------------------------------------------------------
def biot_savart_film_to_film(*, film1_sites, film1_z0, film1_areas, film1_J, film2_sites,
                             film2_z0, block: int = 2048) -> np.ndarray:
    """solver/solve.py:28-73 -- H_i = sum_j (1/4pi) a_j (Jx_j dy - Jy_j dx) r^-3,
    dx = x_i(target) - x_j(source); output dtype = dtype of ``film1_J``."""
    one_over_4pi = 1 / (4 * np.pi)
    dz2 = (film2_z0 - film1_z0) ** 2
    m = film2_sites.shape[0]
    out = np.empty(m, dtype=film1_J.dtype)
    xs, ys = film1_sites[:, 0], film1_sites[:, 1]
    aJx = film1_areas * film1_J[:, 0]
    aJy = film1_areas * film1_J[:, 1]
    for i0 in range(0, m, block):
        i1 = min(m, i0 + block)
        dx = film2_sites[i0:i1, 0, None] - xs[None, :]
        dy = film2_sites[i0:i1, 1, None] - ys[None, :]
        r3 = (dx * dx + dy * dy + dz2) ** (-1.5)
        out[i0:i1] = one_over_4pi * np.sum((aJx * dy - aJy * dx) * r3, axis=1)
    return out


def solve(films: Sequence[OracleFilm], applied_field_mT, *, iterations: int = 0,
          circulating_currents: Optional[Dict[str, float]] = None,
          field_conversion: Optional[float] = None,
          biot_savart=None, vortices: Optional[Dict[str, Sequence[Tuple[float, float, float]]]] = None,
          terminal_currents: Optional[Dict[str, Dict[str, float]]] = None) -> List[Dict[str, OracleFilmSolution]]:
    """solver/solve.py:422-547 -- first pass, then ``iterations`` Jacobi rounds.  Returns the
    per-iteration list (length ``iterations + 1``; 1 for a single film, :486-489).

    ``applied_field_mT``: float (uniform field in mT) or callable ``f(x, y, z)``.
    ``vortices``: ``{film: [(x, y, nPhi0), ...]}`` -- the trapped vortices by film (solver/utils.py:205-231).
    ``terminal_currents``: ``{film: {terminal: current}}`` for films made with ``terminal_masks`` (solver/solve.py:530).
    ``biot_savart``: the film-to-film kernel to use (default: the numpy restatement above; the
    headline-size checks pass the OpenMP C port ``cpu_kernels.biot_savart_film_to_film``, which the
    CPU tests hold against the numpy form).
    """
    pair_field = biot_savart or biot_savart_film_to_film
    conv = field_conversion_mT_to_uA_per_um() if field_conversion is None else field_conversion
    applied = {}
    for f in films:
        x, y = f.mesh.sites[:, 0], f.mesh.sites[:, 1]
        z = f.z0 * np.ones(len(x))
        val = applied_field_mT(x, y, z) if callable(applied_field_mT) else applied_field_mT * np.ones_like(x)
        applied[f.name] = np.squeeze(val * conv).astype(f.dtype, copy=False)  # :426-430

    def one_pass(other):
        return {
            f.name: solve_film(
                f, applied[f.name], field_conversion=conv,
                circulating_currents=circulating_currents,
                field_from_other_films=None if other is None else other[f.name],
                vortices=(vortices or {}).get(f.name, ()),
                terminal_currents=(terminal_currents or {}).get(f.name),
            )
            for f in films
        }

    sols = one_pass(None)
    out = [sols]
    if len(films) < 2 or iterations < 1:
        return out
    for _ in range(iterations):
        other = {f.name: np.zeros(len(f.mesh.sites), dtype=f.dtype) for f in films}
        for src, tgt in itertools.product(films, repeat=2):  # :499-515
            if src is tgt:
                continue
            other[tgt.name] += pair_field(
                film1_sites=src.mesh.sites, film1_z0=src.z0, film1_areas=src.weights,
                film1_J=sols[src.name].current_density, film2_sites=tgt.mesh.sites,
                film2_z0=tgt.z0,
            )
        sols = one_pass(other)  # Jacobi: all films use the previous iterate (:520-536)
        out.append(sols)
    return out


# --------------------------------------------------------------------------------------
# Fluxoid (solution.py:484-563, 278-319) in raw units: field_units * length^2
# --------------------------------------------------------------------------------------
def polygon_fluxoid_raw(film: OracleFilm, sol: OracleFilmSolution, polygon_points: np.ndarray,
                        in_polygon: np.ndarray, in_film_poly: np.ndarray, lambda_func=None) -> Tuple[float, float]:
    """Returns ``(flux_part [mT um^2], int_J [uA um])``.

    ``flux_part = sum_{i in polygon} total_field_i w_i`` (solution.py:535-538);
    ``int_J = trapezoid(Lambda_k * sum_xy(J_k * dl_k))`` over polygon vertices k = 0..N-2 with
    unit spacing (:556-559), J from ``matplotlib.tri.LinearTriInterpolator`` with non-finite /
    out-of-film values set to 0 (:313-315).  The supercurrent part in field*length^2 is
    ``mu_0 * int_J``.  ``in_polygon``: polygon.contains_points(sites); ``in_film_poly``:
    film.contains_points(polygon_points).
    """
    from matplotlib.tri import LinearTriInterpolator, Triangulation

    flux_part = float(np.sum(sol.total_field[in_polygon] * film.mesh.weights[in_polygon]))
    tri = Triangulation(film.mesh.sites[:, 0], film.mesh.sites[:, 1], film.mesh.elements)
    J = sol.current_density
    xv, yv = polygon_points[:, 0], polygon_points[:, 1]
    Jp = np.array([
        LinearTriInterpolator(tri, J[:, 0])(xv, yv).data,
        LinearTriInterpolator(tri, J[:, 1])(xv, yv).data,
    ]).T
    Jp[~in_film_poly] = 0
    Jp[~np.isfinite(Jp).all(axis=1)] = 0
    # solution.py:548-551: a Parameter Lambda is evaluated at the polygon's vertices (``lambda_func(x, y)``)
    Lam = (np.asarray(lambda_func(xv, yv), dtype=float) if lambda_func is not None
           else float(film.Lambda[0]) * np.ones(len(polygon_points)))
    dl = np.diff(polygon_points, axis=0)
    int_J = float(np.trapezoid(Lam[:-1] * np.sum(Jp[:-1] * dl, axis=1)))
    return flux_part, int_J


def polygon_fluxoid_mT_um2(film: OracleFilm, sol: OracleFilmSolution, polygon_points: np.ndarray,
                           film_outline: np.ndarray) -> Tuple[float, float]:
    """``Solution.polygon_fluxoid(polygon, film=...)`` (solution.py:484-563) of one oracle film solution as
    ``(flux_part, supercurrent_part)``, both in mT um^2 (``units="mT * um**2"``): the vertex test of the flux part
    is ``Polygon.contains_points`` = matplotlib ``Path.contains_points`` with radius 0 (device/polygon.py:159), the
    in-film test of the polygon vertices the same on the film's outline (solution.py:313-315).  ``polygon_points``:
    the closed, counter-clockwise ring as ``Polygon.points`` holds it."""
    from matplotlib.path import Path

    polygon_points = np.asarray(polygon_points, dtype=np.float64)
    in_polygon = Path(polygon_points, closed=True).contains_points(film.mesh.sites, radius=0)
    in_film_poly = Path(np.asarray(film_outline, dtype=np.float64), closed=True).contains_points(polygon_points, radius=0)
    flux_raw, int_J = polygon_fluxoid_raw(film, sol, polygon_points, in_polygon, in_film_poly)
    return flux_raw, MU_0 * int_J * 1e-12 / (1e-3 * 1e-12)   # mu_0 [uA um] -> mT um^2


def fluxoid_in_Phi0(flux_part_mT_um2: float, int_J_uA_um: float) -> Tuple[float, float]:
    """Unit conversion of the two parts to Phi_0 (solution.py:538, 560-561)."""
    flux = flux_part_mT_um2 * 1e-3 * 1e-12 / PHI_0
    sc = MU_0 * int_J_uA_um * 1e-6 * 1e-6 / PHI_0
    return flux, sc


# ---------------------------------------------------------------------------------------
# Field of a current sheet at arbitrary points (sources/current.py:13-199)
# ---------------------------------------------------------------------------------------
def biot_savart_2d(x, y, z, *, positions, current_densities, z0=0.0, areas, length_units_to_m=1e-6,
                   current_density_units_to_A_per_m=1.0, vector=True) -> np.ndarray:
    """sources/current.py:113-199 with the two numba kernels (:13-57, :60-110) as one vectorised
    expression: everything is converted to metres and A/m first, exactly as the wrapper does
    (:166-180), then  pref = mu_0/(4 pi) a_k |r - r_k|^-3  and
    B = sum_k pref (Jy dz, -Jx dz, Jx dy - Jy dx).  Returns tesla, (n, 3) or the z component (n,)."""
    x, y, z = np.atleast_1d(x, y, z)
    if z.shape[0] == 1:
        z = z * np.ones_like(x)
    ev = np.array([x, y, z], dtype=float).T * length_units_to_m
    pos = np.atleast_2d(positions) * length_units_to_m
    J = np.atleast_2d(current_densities) * current_density_units_to_A_per_m
    a = np.asarray(areas, dtype=float) * length_units_to_m**2
    zs = z0 * length_units_to_m
    dx = ev[:, 0, None] - pos[None, :, 0]
    dy = ev[:, 1, None] - pos[None, :, 1]
    dz = (ev[:, 2] - zs)[:, None] * np.ones_like(dx)
    # NB: this module of the reference takes mu_0 from scipy.constants (sources/current.py:5), not
    # from pint like the solver (CODATA 2018 there); with scipy >= 1.15 the two differ by 6.8e-10.
    from scipy.constants import mu_0 as mu_0_scipy

    pref = (mu_0_scipy / (4 * np.pi)) * a[None, :] * (dx * dx + dy * dy + dz * dz) ** (-1.5)
    Jx, Jy = J[None, :, 0], J[None, :, 1]
    Bz = np.sum(pref * Jx * dy, axis=1) - np.sum(pref * Jy * dx, axis=1)
    if not vector:
        return Bz
    return np.stack([np.sum(pref * Jy * dz, axis=1), -np.sum(pref * Jx * dz, axis=1), Bz], axis=1)


def vector_potential(positions_xyz, *, sites, z0, areas, J, current_to_A=1e-6, out_T_m_to=1e3 * 1e6) -> np.ndarray:
    """solution.py:900-931 for one film: ``A = mu_0 / (4 pi) sum_j a_j J_j / rho_ij``, rho including dz;
    J in uA/um and lengths in um give a current in uA; result converted T m -> mT um by default.
    Returns ``(m, 3)`` with ``A_z = 0``."""
    positions_xyz = np.atleast_2d(np.asarray(positions_xyz, dtype=float))
    d2 = ((positions_xyz[:, None, :2] - sites[None, :, :]) ** 2).sum(axis=2)
    rho = np.sqrt(d2 + (positions_xyz[:, 2, None] - z0) ** 2)[:, :, None]
    Axy = np.einsum("ijk, j -> ik", J / rho, areas)
    A = np.concatenate([Axy, np.zeros_like(Axy[:, :1])], axis=1)
    return MU_0 / (4 * np.pi) * A * current_to_A * out_T_m_to


def polygon_flux_raw(total_field, vertex_areas, in_polygon) -> float:
    """solution.py:470-478 in raw units (field_units * length_units^2): ``sum_{i in polygon} B_i a_i``."""
    ix = np.where(in_polygon)[0]
    return float(np.einsum("i, i -> ", total_field[ix], vertex_areas[ix]))
