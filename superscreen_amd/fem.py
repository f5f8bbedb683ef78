"""Host-side construction of the sparse mesh operators (vectorised numpy/scipy.sparse).

These are the O(n) set-up steps in front of the GPU hot path (SURVEY.md section 8a, rows
a4-a7): vertex areas, the half-cotangent Laplacian and the triangle / vertex gradient
operators.  The reference builds them with Python loops and LIL fancy indexing
(``device/utils.py:251-273``, ``fem.py:165-224, 259-296, 299-347, 350-402``); here every
operator is assembled in one shot from COO triplets.  The dense n x n operators (Q, A) are
never formed on the host -- they are generated on the GPU (``kernels.q_assemble`` /
``kernels.system_assemble``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import scipy.sparse as sp


def _corners(points: np.ndarray, triangles: np.ndarray):
    p0, p1, p2 = (points[triangles[:, k]] for k in range(3))
    return p0, p1, p2


def triangle_areas(points: np.ndarray, triangles: np.ndarray) -> np.ndarray:
    """Signed triangle areas, positive for CCW (``device/utils.py:230-248``)."""
    p0, p1, p2 = _corners(points, triangles)
    a, b = p2 - p1, p0 - p2
    return 0.5 * (a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0])


def vertex_areas(points: np.ndarray, triangles: np.ndarray,
                 tri_areas: Optional[np.ndarray] = None) -> np.ndarray:
    """``w_i`` = one third of the area of the triangles around vertex i
    (``device/utils.py:251-273``)."""
    if tri_areas is None:
        tri_areas = triangle_areas(points, triangles)
    return np.bincount(triangles.ravel(), weights=np.repeat(tri_areas / 3, 3), minlength=len(points))


def boundary_indices(triangles: np.ndarray) -> np.ndarray:
    """Sorted indices of vertices on edges that belong to a single triangle
    (``device/mesh.py:158-170`` + ``device/utils.py:139-152``)."""
    e = np.sort(np.concatenate([triangles[:, [0, 1]], triangles[:, [1, 2]], triangles[:, [2, 0]]]), axis=1)
    n = int(triangles.max()) + 1
    key = e[:, 0].astype(np.int64) * n + e[:, 1]
    uniq, counts = np.unique(key, return_counts=True)
    once = uniq[counts == 1]
    return np.unique(np.concatenate([once // n, once % n]))


def boundary_vertices(points: np.ndarray, triangles: np.ndarray) -> np.ndarray:
    """Boundary vertex indices ordered counter-clockwise (``device/utils.py:205-226``).

    The reference polygonises the boundary edges with shapely; here the directed edges of the
    (counter-clockwise) triangles that have no twin are chained into the outer loop, starting at
    its lowest vertex index.  (The reference's start vertex is whatever shapely returns; the
    callers only rely on the cyclic order, ``Device.boundary_vertices`` re-rolls the loop so that
    it does not wrap inside a terminal.)"""
    tri = np.asarray(triangles, dtype=np.int64)
    p = points[tri]
    area2 = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 2, 0] - p[:, 0, 0]) * (p[:, 1, 1] - p[:, 0, 1])
    tri = np.where(area2[:, None] < 0, tri[:, [0, 2, 1]], tri)
    a = np.concatenate([tri[:, 0], tri[:, 1], tri[:, 2]])
    b = np.concatenate([tri[:, 1], tri[:, 2], tri[:, 0]])
    n = int(tri.max()) + 1
    twins = set((b * n + a).tolist())
    nxt = {int(u): int(v) for u, v in zip(a, b) if int(u) * n + int(v) not in twins}
    if not nxt:
        return np.zeros(0, dtype=np.int64)
    start = min(nxt)
    loop = [start]
    while nxt[loop[-1]] != start:
        loop.append(nxt[loop[-1]])
        if len(loop) > len(nxt):
            raise ValueError("The mesh boundary is not a single closed loop.")
    if len(loop) != len(nxt):
        raise ValueError("The mesh boundary is not a single closed loop.")
    return np.asarray(loop, dtype=np.int64)


def _angle(u: np.ndarray, v: np.ndarray) -> np.ndarray:
    """Angle between 2-D vectors; the reference takes ``arccos`` of the normalised dot product
    (``fem.py:188-224, 393-399``) -- same convention here so results agree to rounding."""
    c = np.sum(u * v, axis=1) / (np.linalg.norm(u, axis=1) * np.linalg.norm(v, axis=1))
    return np.arccos(c)


def laplace_operator(points: np.ndarray, triangles: np.ndarray,
                     masses: Optional[np.ndarray] = None) -> sp.csr_array:
    """Half-cotangent Laplacian ``diag(1/w) @ (W - diag(rowsum W))`` (``fem.py:259-296`` with
    ``weights_half_cotangent``, ``fem.py:165-224``): edge (i, j) gets ``1/2 cot`` of each angle
    opposite to it."""
    n = len(points)
    if masses is None:
        masses = vertex_areas(points, triangles)
    p = _corners(points, triangles)
    rows, cols, vals = [], [], []
    for c in range(3):
        a, b = (c + 1) % 3, (c + 2) % 3
        half_cot = 0.5 / np.tan(_angle(p[a] - p[c], p[b] - p[c]))
        i, j = triangles[:, a], triangles[:, b]
        rows += [i, j]
        cols += [j, i]
        vals += [half_cot, half_cot]
    W = sp.coo_array((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                     shape=(n, n)).tocsr()
    rowsum = np.asarray(W.sum(axis=1)).ravel()
    L = W - sp.diags_array(rowsum, format="csr")
    lap = (sp.diags_array(1.0 / masses, format="csr") @ L).tocsr()
    lap.sort_indices()
    return lap


def gradient_triangles(points: np.ndarray, triangles: np.ndarray,
                       areas: Optional[np.ndarray] = None) -> Tuple[sp.csr_array, sp.csr_array]:
    """Per-triangle gradient of a piecewise-linear field (``fem.py:299-347``): the coefficient of
    vertex k is the opposite edge rotated by -90 degrees over twice the area."""
    if areas is None:
        areas = triangle_areas(points, triangles)
    m, n = len(triangles), len(points)
    p0, p1, p2 = _corners(points, triangles)
    opp = np.stack([p1 - p2, p2 - p0, p0 - p1], axis=1)  # edge opposite to corner k
    gx = opp[:, :, 1] / (2 * areas[:, None])
    gy = -opp[:, :, 0] / (2 * areas[:, None])
    row = np.repeat(np.arange(m), 3)
    col = triangles.ravel()
    Gx = sp.csr_array((gx.ravel(), (row, col)), shape=(m, n))
    Gy = sp.csr_array((gy.ravel(), (row, col)), shape=(m, n))
    return Gx, Gy


def gradient_vertices(points: np.ndarray, triangles: np.ndarray,
                      gradient_tri: Optional[Tuple[sp.csr_array, sp.csr_array]] = None
                      ) -> Tuple[sp.csr_array, sp.csr_array]:
    """Vertex gradient = weighted mean of the gradients of the adjacent triangles
    (``fem.py:350-402``).  The weight of triangle t is its angle at ITS FIRST vertex
    (``vec1 = p[t1] - p[t0]``, ``vec2 = p[t2] - p[t0]``, :393-399 -- independent of the vertex
    being averaged), normalised over the triangles adjacent to the vertex; kept as is because
    J feeds the inter-film coupling."""
    if gradient_tri is None:
        gradient_tri = gradient_triangles(points, triangles)
    Gx, Gy = gradient_tri
    m, n = len(triangles), len(points)
    p0, p1, p2 = _corners(points, triangles)
    ang = _angle(p1 - p0, p2 - p0)
    vert = triangles.ravel()
    tri = np.repeat(np.arange(m), 3)
    norm = np.bincount(vert, weights=ang[tri], minlength=n)
    Wt = sp.csr_array((ang[tri] / norm[vert], (vert, tri)), shape=(n, m))
    gx, gy = (Wt @ Gx).tocsr(), (Wt @ Gy).tocsr()
    gx.sort_indices()
    gy.sort_indices()
    return gx, gy


def shared_pattern(gx: sp.csr_array, gy: sp.csr_array):
    """One CSR pattern for both vertex-gradient operators (what ``ssa_current_density``
    consumes): returns ``(indptr, indices, gx_values, gy_values)`` on the union pattern."""
    n = gx.shape[0]
    pattern = (abs(gx) + abs(gy)).tocsr()
    pattern.sort_indices()
    indptr = pattern.indptr.astype(np.int64)
    indices = pattern.indices.astype(np.int64)
    rows = np.repeat(np.arange(n), np.diff(indptr))

    def on_pattern(m):
        m = m.tocsr()
        m.sort_indices()
        out = np.zeros(len(indices))
        key_p = rows * m.shape[1] + indices
        mrows = np.repeat(np.arange(n), np.diff(m.indptr))
        key_m = mrows * m.shape[1] + m.indices
        pos = np.searchsorted(key_p, key_m)
        out[pos] = m.data
        return out

    return indptr, indices, on_pattern(gx), on_pattern(gy)
