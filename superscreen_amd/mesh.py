"""``Mesh`` and ``MeshOperators`` (reference: ``device/mesh.py:17-160, 326-458``).

Data layout kept from the reference: ``sites (n, 2) float64``,
``elements (m, 3) int64`` counter-clockwise, ``boundary_indices int64``, ``vertex_areas``,
``triangle_areas``; ``mesh.operators.{weights, Q, gradient_x, gradient_y, gradient_tri_x,
gradient_tri_y, laplacian}``.

Difference by design: the dense kernel matrix ``Q`` (n^2 * 8 bytes: 20 GB at n = 50k) is NOT
built eagerly on the host as ``Mesh.__init__`` does in the reference
(``device/mesh.py:59-60, 375``).  It is generated on the GPU by ``ssa_q_assemble`` where the
solver needs it; ``MeshOperators.Q`` materialises a host copy only when somebody asks for it.
"""
from __future__ import annotations

import itertools
from typing import Optional

import numpy as np
import scipy.sparse as sp

from . import fem


class MeshOperators:
    """Operators of a mesh (``device/mesh.py:326-394``)."""

    def __init__(self, *, sites: np.ndarray, weights: np.ndarray, gradient_x: sp.csr_array,
                 gradient_y: sp.csr_array, gradient_tri_x: sp.csr_array,
                 gradient_tri_y: sp.csr_array, laplacian: sp.csr_array):
        self._sites = sites
        self.weights = weights
        self.gradient_x = gradient_x
        self.gradient_y = gradient_y
        self.gradient_tri_x = gradient_tri_x
        self.gradient_tri_y = gradient_tri_y
        self.laplacian = laplacian
        self._Q: Optional[np.ndarray] = None
        self._C: Optional[np.ndarray] = None
        self._device_cache = {}  # torch device -> DeviceMeshData (see solver.py)

    @staticmethod
    def from_mesh(mesh: "Mesh") -> "MeshOperators":
        """``MeshOperators.from_mesh`` (``device/mesh.py:362-394``) minus the dense Q."""
        sites, elements = mesh.sites, mesh.elements
        Gx, Gy = fem.gradient_triangles(sites, elements, mesh.triangle_areas)
        gx, gy = fem.gradient_vertices(sites, elements, (Gx, Gy))
        return MeshOperators(
            sites=sites,
            weights=mesh.vertex_areas,
            gradient_x=gx,
            gradient_y=gy,
            gradient_tri_x=Gx,
            gradient_tri_y=Gy,
            laplacian=fem.laplace_operator(sites, elements, mesh.vertex_areas),
        )

    @staticmethod
    def C_vector(points: np.ndarray) -> np.ndarray:
        """Edge vector ``C_i = (1/4pi) sum_{p,q=+-1} sqrt((a - p x_i)^-2 + (b - q y_i)^-2)``
        (``device/mesh.py:401-432``); coordinates are centred on their MEAN (:421-422) and
        infinities are replaced by 1e30 before the division by 4 pi (:430-431)."""
        x = points[:, 0] - points[:, 0].mean()
        y = points[:, 1] - points[:, 1].mean()
        a = np.ptp(x) / 2
        b = np.ptp(y) / 2
        with np.errstate(divide="ignore"):
            C = sum(np.sqrt((a - p * x) ** (-2) + (b - q * y) ** (-2))
                    for p, q in itertools.product((-1, 1), repeat=2))
        C[np.isinf(C)] = 1e30
        C /= 4 * np.pi
        return C

    @property
    def C(self) -> np.ndarray:
        if self._C is None:
            self._C = MeshOperators.C_vector(self._sites)
        return self._C

    @staticmethod
    def Q_matrix(points: np.ndarray, weights: np.ndarray) -> np.ndarray:
        """Kernel matrix (``device/mesh.py:435-458``), assembled on the GPU (float64) and copied
        to the host.  Fails loudly without a GPU / the HIP library -- there is no CPU path."""
        import torch

        from . import _hip, kernels

        _hip.require_gpu()
        xy = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64)).cuda()
        w = torch.from_numpy(np.ascontiguousarray(weights, dtype=np.float64)).cuda()
        C = torch.from_numpy(MeshOperators.C_vector(points)).cuda()
        Q, _ = kernels.q_assemble(xy, w, C, "float64")
        n = len(points)
        return Q[:, :n].cpu().numpy()

    @property
    def Q(self) -> np.ndarray:
        """Host copy of the kernel matrix (lazy; see the module docstring)."""
        if self._Q is None:
            self._Q = MeshOperators.Q_matrix(self._sites, self.weights)
        return self._Q


class Mesh:
    """A triangular mesh of a simply- or multiply-connected polygon
    (``device/mesh.py:17-60``)."""

    def __init__(self, sites, elements, boundary_indices, vertex_areas, triangle_areas,
                 build_operators: bool = True):
        self.sites = np.ascontiguousarray(np.asarray(sites, dtype=np.float64).squeeze())
        self.elements = np.ascontiguousarray(np.asarray(elements, dtype=np.int64))
        self.triangle_centroids = self.sites[self.elements].mean(axis=1)
        self.boundary_indices = np.asarray(boundary_indices, dtype=np.int64)
        self.vertex_areas = np.asarray(vertex_areas)
        self.triangle_areas = np.asarray(triangle_areas)
        self.operators: Optional[MeshOperators] = None
        self._triangulation = None
        if build_operators:
            self.operators = MeshOperators.from_mesh(self)

    @property
    def triangulation(self):
        """Matplotlib triangulation of the mesh (``device/mesh.py:62-69``)."""
        if self._triangulation is None:
            from matplotlib.tri import Triangulation

            self._triangulation = Triangulation(self.sites[:, 0], self.sites[:, 1], self.elements)
        return self._triangulation

    def closest_site(self, xy) -> int:
        return int(np.argmin(np.linalg.norm(self.sites - np.atleast_2d(xy), axis=1)))

    @staticmethod
    def from_triangulation(sites, elements, build_operators: bool = True) -> "Mesh":
        """``Mesh.from_triangulation`` (``device/mesh.py:111-155``)."""
        sites = np.asarray(sites).squeeze()
        elements = np.asarray(elements).squeeze()
        if sites.ndim != 2 or sites.shape[1] != 2:
            raise ValueError(f"The site coordinates must have shape (n, 2), got {sites.shape!r}")
        if elements.ndim != 2 or elements.shape[1] != 3:
            raise ValueError(f"The elements must have shape (m, 3), got {elements.shape!r}.")
        elements = elements.astype(np.int64)
        sites = sites.astype(np.float64)
        tri_areas = fem.triangle_areas(sites, elements)
        return Mesh(
            sites=sites,
            elements=elements,
            boundary_indices=fem.boundary_indices(elements),
            vertex_areas=fem.vertex_areas(sites, elements, tri_areas),
            triangle_areas=tri_areas,
            build_operators=build_operators,
        )

    @staticmethod
    def find_boundary_indices(elements: np.ndarray) -> np.ndarray:
        return fem.boundary_indices(np.asarray(elements))

    def to_hdf5(self, h5group, compress: bool = True) -> None:
        """``device/mesh.py:250-264``: sites and elements; with ``compress=False`` also the derived
        arrays (the reference's edge mesh is not kept by this package and is not written)."""
        h5group["sites"] = self.sites
        h5group["elements"] = self.elements
        if not compress:
            h5group["triangle_centroids"] = self.triangle_centroids
            h5group["boundary_indices"] = self.boundary_indices
            h5group["vertex_areas"] = self.vertex_areas
            h5group["triangle_areas"] = self.triangle_areas

    @staticmethod
    def from_hdf5(h5group) -> "Mesh":
        """``device/mesh.py:266-293``: the derived arrays are taken from the file when all of them are
        there, otherwise recomputed from the triangulation."""
        if not ("sites" in h5group and "elements" in h5group):
            raise IOError("Could not load mesh due to missing data.")
        sites = np.array(h5group["sites"]).squeeze()
        elements = np.array(h5group["elements"], dtype=np.int64)
        if all(key in h5group for key in ("boundary_indices", "vertex_areas", "triangle_areas")):
            return Mesh(sites, elements, np.array(h5group["boundary_indices"], dtype=np.int64),
                        np.array(h5group["vertex_areas"]), np.array(h5group["triangle_areas"]))
        return Mesh.from_triangulation(sites, elements)

    def copy(self) -> "Mesh":
        return Mesh(self.sites.copy(), self.elements.copy(), self.boundary_indices.copy(),
                    self.vertex_areas.copy(), self.triangle_areas.copy(),
                    build_operators=self.operators is not None)
