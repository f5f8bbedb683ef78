"""The solver hot path: ``factorize_model`` and ``solve`` on an MI355X.

Host orchestration mirrors the reference line by line where behaviour is observable
(``solver/solve.py:223-549``, ``solver/solve_film.py:151-282, 440-574``,
``solver/utils.py:19-132, 234-324``); every numerical step runs in the HIP library
(``include/superscreen_hip.h``) on device-resident data:

  make_film_info            host index logic (point-in-polygon, setdiff1d)      utils.py:271-304
  factorize_linear_systems  ssa_q_assemble (diagonal) -> ssa_system_assemble    solve_film.py:209-218, 275
                            (-A generated straight into the LU buffer)
                            -> ssa_lu_factor                                     solve_film.py:279
  solve_film                ssa_index_add_scalar / ssa_gemv (holes)             solve_film.py:498-503
                            ssa_film_rhs -> ssa_lu_solve -> ssa_scatter_add      :526-531
                            ssa_current_density                                  :556
                            ssa_self_field (matrix-free) or ssa_gemv on a stored Q  :565
  solve                     ssa_biot_savart per ordered film pair, Jacobi loop   solve.py:491-536

Memory (per film, n vertices, n_i unknowns, s = sizeof(solve dtype)): LU n_i^2 s (13.7 GB at
n = 50k in f64), hole systems n * k_h * s, optional stored Q n^2 s; the reference additionally
keeps A, a float64 Q and a DENSE Laplacian (solver/utils.py:290-292), about 5 n^2 words.

Persistence (``to_hdf5`` / ``from_hdf5`` / ``save_path``): superscreen_amd.io.  Transport currents through terminals
(solve_film.py:308-437, 505-524, 557-562) reuse the film factorization where the reference factors
the same matrix again, and run their all-pairs sums through ``ssa_sheet_field``.  Vortices (solve_film.py:541-554) are one extra right-hand side per vortex through the
existing factorization; a film with Lambda(x, y) (grad(Lambda) term, :181-185) goes through the LU
route because diag(w) A is then no longer symmetric.
"""
from __future__ import annotations

import copy as _copy
import itertools
import os
import logging
import numbers
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import scipy.sparse as sp

from . import fem
from .device import Device
from .parameter import Constant
from .geometry import close_curve, path_vectors
from .solution import FilmSolution, Solution, Vortex
from .sources import ConstantField
from .units import current_to_float, field_conversion_factor, vortex_flux

logger = logging.getLogger("solve")


# ---------------------------------------------------------------------------------------
# Film bookkeeping (host)
# ---------------------------------------------------------------------------------------
class LambdaInfo:
    """Effective penetration depth of a film on its mesh (``solver/utils.py:19-58``)."""

    def __init__(self, *, film: str, Lambda: np.ndarray, london_lambda: Optional[np.ndarray] = None,
                 thickness: Optional[float] = None):
        self.film = film
        self.Lambda = Lambda
        self.london_lambda = london_lambda
        self.thickness = thickness
        self.inhomogeneous = bool(
            np.ptp(self.Lambda) / max(np.min(np.abs(self.Lambda)), np.finfo(float).eps) > 1e-6
        )
        if self.inhomogeneous:
            logger.info(f"Inhomogeneous Lambda in film {self.film!r}, which violates the "
                        "assumptions of the London model. Results may not be reliable.")
        if self.london_lambda is not None:
            assert self.thickness is not None
            assert np.allclose(self.Lambda, self.london_lambda ** 2 / self.thickness)
        if np.any(self.Lambda < 0):
            raise ValueError(f"Negative Lambda in film {film!r}.")


@dataclass
class FilmInfo:
    """Per-film data required by the solver (``solver/utils.py:96-132``).  ``kernel`` is not
    stored: the dense Q lives (optionally) on the GPU, see ``FilmDeviceData``."""

    name: str
    layer: str
    lambda_info: LambdaInfo
    vortices: Tuple[Vortex, ...]
    interior_indices: np.ndarray
    boundary_indices: np.ndarray
    hole_indices: Dict[str, np.ndarray]
    in_hole: np.ndarray
    circulating_currents: Dict[str, float]
    weights: np.ndarray
    laplacian: sp.csr_array
    gradient: Optional[np.ndarray] = None
    terminal_currents: Optional[Dict[str, float]] = None
    z0: float = 0.0


def get_holes_and_vortices_by_film(device: Device, vortices: Sequence[Vortex]):
    """``solver/utils.py:212-231`` (incl. its error conventions)."""
    vortices_by_film = {name: [] for name in device.films}
    holes_by_film = device.holes_by_film()
    for vortex in vortices:
        if not isinstance(vortex, Vortex):
            raise TypeError(f"Expected a Vortex, but got {type(vortex)}.")
        if not device.films[vortex.film].contains_points((vortex.x, vortex.y)).all():
            raise ValueError(f"Vortex {vortex!r} is not located in film {vortex.film!r}.")
        for hole in holes_by_film[vortex.film]:
            if hole.contains_points((vortex.x, vortex.y)).all():
                raise ValueError(f"Vortex {vortex} is located in hole {hole.name!r}.")
        vortices_by_film[vortex.film].append(vortex)
    return holes_by_film, vortices_by_film


def _sites_in_polygon(mesh, polygon) -> np.ndarray:
    """``polygon.contains_points(mesh.sites, index=True)`` (``solver/utils.py:271-274, 302``),
    memoised per (mesh, polygon geometry): the point-in-polygon test is pure host work that
    repeated ``factorize_model`` calls on the same device would otherwise redo."""
    cache = mesh.__dict__.setdefault("_contains_cache", {})
    key = hash(polygon.points.tobytes())
    if key not in cache:
        cache[key] = polygon.contains_points(mesh.sites, index=True)
    return cache[key]


def _index_setdiff(a, b, n: int) -> np.ndarray:
    """``np.setdiff1d(a, b)`` for index arrays into ``range(n)`` (sorted, unique) without its sorts: a cold step
    calls this a few times per film before its first kernel can be launched."""
    keep = np.zeros(n, dtype=bool)
    keep[np.asarray(a, dtype=np.int64)] = True
    keep[np.asarray(b, dtype=np.int64)] = False
    return np.flatnonzero(keep)


def make_film_info(*, device: Device, vortices: Sequence[Vortex],
                   circulating_currents: Dict[str, float],
                   terminal_currents: Dict[str, Dict[str, float]]) -> Dict[str, FilmInfo]:
    """``make_film_info`` (``solver/utils.py:234-324``)."""
    dtype = device.solve_dtype
    holes_by_film, vortices_by_film = get_holes_and_vortices_by_film(device, vortices)
    film_info = {}
    for name, film in device.films.items():
        mesh = device.meshes[name]
        layer = device.layers[film.layer]
        london_lambda, d, Lambda = layer.london_lambda, layer.thickness, layer.Lambda
        if isinstance(london_lambda, numbers.Real) and london_lambda <= d:
            logger.info(f"Layer {name!r}: film thickness d = {d:.4f} >= london_lambda = "
                        f"{london_lambda:.4f}; the thin-film assumption may not be valid.")
        if isinstance(Lambda, numbers.Real):
            Lambda = Constant(Lambda)
        Lambda = np.asarray(Lambda(mesh.sites[:, 0], mesh.sites[:, 1]) * np.ones(len(mesh.sites)))
        Lambda = Lambda.astype(dtype, copy=False)[:, np.newaxis]
        if london_lambda is not None:
            if isinstance(london_lambda, numbers.Real):
                london_lambda = Constant(london_lambda)
            london_lambda = np.asarray(
                london_lambda(mesh.sites[:, 0], mesh.sites[:, 1]) * np.ones(len(mesh.sites))
            ).astype(dtype, copy=False)[:, np.newaxis]
        hole_indices = {hole.name: _sites_in_polygon(mesh, hole) for hole in holes_by_film[name]}
        in_hole = np.zeros(len(mesh.sites), dtype=bool)
        if hole_indices:
            in_hole[np.concatenate(list(hole_indices.values()))] = True
        circ = {h: c for h, c in circulating_currents.items() if h in hole_indices}
        lambda_info = LambdaInfo(film=name, Lambda=Lambda, london_lambda=london_lambda,
                                 thickness=layer.thickness)
        if name in device.terminals:
            boundary = device.boundary_vertices(name)  # ordered, solver/utils.py:298-299
        else:
            boundary = mesh.boundary_indices
        interior = _index_setdiff(_sites_in_polygon(mesh, film), boundary, len(mesh.sites))
        film_info[name] = FilmInfo(
            name=name, layer=layer.name, lambda_info=lambda_info,
            vortices=tuple(vortices_by_film[name]), interior_indices=interior,
            boundary_indices=boundary, hole_indices=hole_indices, in_hole=in_hole,
            circulating_currents=circ, terminal_currents=terminal_currents.get(name),
            weights=mesh.operators.weights.astype(dtype, copy=False),
            # (the mesh's own matrix when the dtype already matches: read-only here, as in the reference)
            laplacian=(mesh.operators.laplacian if mesh.operators.laplacian.dtype == dtype
                       else mesh.operators.laplacian.astype(dtype)),
            z0=float(layer.z0),
        )
    return film_info


# ---------------------------------------------------------------------------------------
# Device-resident state
# ---------------------------------------------------------------------------------------
def _h2d(a: np.ndarray, dev):
    """Host array -> device tensor without stalling the host: ``.to(device)`` of a pageable array waits for
    everything queued on the stream before it (the copy is stream-ordered and synchronous), which serialised the
    host behind each film's 0.5 ms row-sum kernel during assembly; a pinned staging buffer and a non-blocking copy
    keep the host ahead of the GPU (the caching host allocator keeps the buffer alive until the copy has run)."""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(a))
    if dev.type != "cuda" or t.numel() == 0 or t.numel() * t.element_size() > (1 << 20):
        return t.to(dev)   # (large arrays: staging them through a fresh pinned buffer costs more than it hides)
    return t.pin_memory().to(dev, non_blocking=True)



class FilmDeviceData:
    """Everything of one film that lives in HBM (torch tensors are plumbing only)."""

    @staticmethod
    def device_geometry(mesh, dtype: np.dtype):
        """Mesh geometry and sparse operators on the current GPU: uploaded once per (mesh, GPU, dtype), they stay
        resident in HBM across factorize_model calls."""
        import torch

        tdt = torch.float64 if dtype == np.float64 else torch.float32
        dev = torch.device("cuda", torch.cuda.current_device())
        ops = mesh.operators
        key = (dev.index, str(dtype))
        geo = ops._device_cache.get(key)
        if geo is None:
            def put(a):
                return _h2d(a, dev)

            lap = ops.laplacian.tocsr()
            lap.sort_indices()
            ptr_, idx_, vx, vy = fem.shared_pattern(ops.gradient_x, ops.gradient_y)
            w = put(ops.weights)
            geo = dict(
                xy=put(mesh.sites), w=w, w_t=w.to(tdt), C=put(ops.C),
                lap=(put(lap.indptr.astype(np.int64)), put(lap.indices.astype(np.int64)), put(lap.data)),
                grad=(put(ptr_), put(idx_), put(vx), put(vy)),
            )
            # The uploads above are non-blocking copies on the CURRENT stream; what is cached here is used by later
            # calls that may run under another stream, so the cache only ever holds completed copies.
            done = torch.cuda.Event()
            done.record()
            done.synchronize()
            ops._device_cache[key] = geo
        return geo

    @staticmethod
    def start_row_sums(mesh, dtype: np.dtype, store_Q: bool) -> None:
        """Launches the kernel-matrix row sums of a film (the first kernel of a cold step, 0.5 ms at 25 000
        vertices) ahead of the host work that prepares its index sets; the constructor picks the result up."""
        from . import kernels

        import threading

        geo = FilmDeviceData.device_geometry(mesh, dtype)
        # parked per calling thread (two threads may factorize devices that share a mesh); whoever started them
        # drops what the constructor has not picked up (drop_row_sums: factorize_model's finally clause)
        geo.setdefault("_row_sums", {})[threading.get_ident()] = \
            (bool(store_Q),) + tuple(kernels.q_assemble(geo["xy"], geo["w"], geo["C"], dtype, want_Q=store_Q))

    @staticmethod
    def drop_row_sums(mesh, dtype: np.dtype) -> None:
        """Forgets row sums started by this thread that no constructor picked up (an exception on the way, a film
        another rank owns): they may hold an n x n matrix."""
        import threading

        geo = FilmDeviceData.device_geometry(mesh, dtype)
        geo.get("_row_sums", {}).pop(threading.get_ident(), None)

    def __init__(self, info: FilmInfo, mesh, dtype: np.dtype, store_Q: bool, geometry_only: bool = False):
        import torch

        from . import kernels

        tdt = torch.float64 if dtype == np.float64 else torch.float32
        dev = torch.device("cuda", torch.cuda.current_device())
        ops = mesh.operators

        def put(a):
            return _h2d(a, dev)

        geo = FilmDeviceData.device_geometry(mesh, dtype)
        self.n = len(mesh.sites)
        self.dtype, self.tdtype, self.device = dtype, tdt, dev
        self.xy, self.w, self.w_t = geo["xy"], geo["w"], geo["w_t"]  # w: f64 geometry; w_t: solve dtype
        self.lap, self.grad = geo["lap"], geo["grad"]
        self.Lambda = put(info.lambda_info.Lambda[:, 0].astype(np.float64))
        self._geo = geo
        # Q_ii needs the full row sums over all n vertices: one all-pairs pass, no n^2 output
        # unless the dense Q is wanted for the self-field GEMV.  (Launched before the host work below: the row sums
        # are the first kernel of a cold step.)
        import threading

        started = geo.get("_row_sums", {}).pop(threading.get_ident(), None)   # (FilmDeviceData.start_row_sums)
        if geometry_only:  # a film owned by another rank: only a coupling source / target geometry
            self.Q = self.qdiag = None
        elif started is not None and started[0] == bool(store_Q):
            self.Q, self.qdiag = started[1], started[2]
        else:
            self.Q, self.qdiag = kernels.q_assemble(self.xy, self.w, geo["C"], dtype, want_Q=store_Q)
        # Index range of the vertices that can carry a sheet current: g lives on the interior and hole
        # vertices (plus the boundary of a film with terminals), J = curl(g z) on those and their mesh
        # neighbours.  Vertices outside the range are exact zeros in every coupling sum and are skipped
        # (with the film vertices numbered first, as a buffered mesh usually is, that is the vacuum ring).
        support = np.zeros(self.n, dtype=bool)
        support[info.interior_indices] = True
        for hole_ix in info.hole_indices.values():
            support[hole_ix] = True
        if info.terminal_currents is not None:
            support[info.boundary_indices] = True
        pattern = geo.get("pattern")
        if pattern is None:  # mesh-only (1-3 ms of host time per call at 25 000 vertices): cached with the geometry
            pattern = geo["pattern"] = (abs(ops.gradient_x) + abs(ops.gradient_y)).tocsr()
        carries = support | ((pattern @ support.astype(np.float64)) > 0)
        self.src_range = (int(np.argmax(carries)), int(self.n - np.argmax(carries[::-1]))) if carries.any() \
            else (0, self.n)

    def triangle_data(self, mesh):
        """Per-triangle operators of films with terminals (centroids, areas, triangle gradient CSR
        with a shared pattern), uploaded on first use and cached with the mesh geometry."""
        import torch

        geo = self._geo
        if "tri" not in geo:
            ops = mesh.operators
            dev = self.device

            def put(a, dt=np.float64):
                return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)

            ptr_, idx_, vx, vy = fem.shared_pattern(ops.gradient_tri_x, ops.gradient_tri_y)
            centroids = mesh.sites[mesh.elements].mean(axis=1)
            geo["tri"] = dict(
                grad=(put(ptr_, np.int64), put(idx_, np.int64), put(vx), put(vy)),
                centroids=put(centroids), areas=put(mesh.triangle_areas),
                sites3=put(np.column_stack([mesh.sites, np.zeros(len(mesh.sites))])),
            )
        return geo["tri"]


@dataclass
class LinearSystem:
    """The linear system of a film or hole (``solver/solve_film.py:19-35``).

    ``A`` and ``lu_piv`` are host views produced on demand from the device-resident data;
    ``lu_piv = (lu, piv)`` follows ``scipy.linalg.lu_factor`` conventions."""

    indices: np.ndarray
    grad_Lambda_term: Union[float, np.ndarray] = 0.0
    _assemble: Optional[Callable[[], "np.ndarray"]] = None   # -> host A
    factors: Optional[object] = None                         # kernels.LUFactors
    chol: Optional[object] = None                            # kernels.CholFactors of diag(w) A
    neg_w_device: Optional[object] = None                    # -w[indices]: rhs scaling of the Cholesky route
    A_device: Optional[object] = None                        # hole systems: [n, ld] tensor
    indices_device: Optional[object] = None
    rhs_indices_device: Optional[object] = None              # indices[perm] (LU row order)
    exterior_device: Optional[object] = None                 # mesh rows that are not unknowns (lazy)
    _lu_factorize: Optional[Callable[[], object]] = None     # builds LUFactors on demand
    _A_host: Optional[np.ndarray] = None

    @property
    def A(self) -> np.ndarray:
        if self._A_host is None:
            self._A_host = self._assemble()
        return self._A_host

    @property
    def lu_piv(self) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        """``(lu, piv)`` of ``-A`` as ``scipy.linalg.lu_factor`` returns them.  When the film was
        factorized through the Cholesky route the LU is computed here, on demand."""
        if self.factors is None:
            if self._lu_factorize is None:
                return None
            self.factors = self._lu_factorize()
        f = self.factors
        return f.lu[:f.n, :f.n].cpu().numpy(), f.ipiv[:f.n].cpu().numpy()


@dataclass
class TerminalSystems:
    """The linear systems behind the transport-current stream function of one film
    (``solver/solve_film.py:80-148``).  ``film_without_boundary_or_holes`` has the same unknowns as
    the film's own system and IS that object here (the reference factors the matrix twice); without
    holes ``film_without_boundary`` is the same object too."""

    film: str
    boundary: LinearSystem
    holes: Dict[str, LinearSystem]
    film_without_boundary: Optional[LinearSystem] = None
    film_without_boundary_or_holes: Optional[LinearSystem] = None


def factorize_linear_systems(device: Device, film_info_dict: Dict[str, FilmInfo], *,
                             store_Q: bool = False, method: str = "auto",
                             owned: Optional[Sequence[str]] = None, solve_block: int = 4096):
    """``factorize_linear_systems`` (``solver/solve_film.py:151-282``) on the GPU.
    Returns ``(film_systems, hole_systems, terminal_systems, film_data)``.

    ``method``: ``"lu"`` factors ``-A`` like the reference (LAPACK getrf semantics);
    ``"cholesky"`` / ``"auto"`` factor the symmetric positive definite ``diag(w) A`` instead
    (half the flops, same solution to rounding), ``"auto"`` falling back to LU if a pivot is
    not positive.  ``"mixed"`` (float64 devices, uniform Lambda, no terminals): the Cholesky factor of ``diag(w) A`` in
    FLOAT32 -- half the factorization time and half the bytes per triangular solve -- and every solve refined in float64
    against the matrix-free ``diag(w) A`` (``MIXED_REFINEMENT_SWEEPS`` sweeps of residual + correction,
    :func:`_mixed_solve`)."""
    if method not in ("auto", "cholesky", "lu", "mixed"):
        raise ValueError(f"Unknown factorization method {method!r}.")
    if method == "mixed":
        if np.dtype(device.solve_dtype) != np.float64:
            raise ValueError("method='mixed' refines a float32 factorization to float64: it needs solve_dtype='float64'.")
        if device.terminals:
            raise NotImplementedError("method='mixed' does not handle films with terminals.")
    import torch

    from . import _hip, kernels

    _hip.require_gpu()
    dtype = device.solve_dtype
    film_systems, hole_systems, film_data, terminal_systems = {}, {}, {}, {}
    pending = []
    # every film's vertex data first: its kernel-matrix row sums (an all-pairs kernel, 0.5 ms at 25 000 vertices) then
    # run on the GPU while the host prepares the index sets of the film before it
    for name, info in film_info_dict.items():
        mine = owned is None or name in owned
        # (another rank's film, parallel.FilmPlacement: geometry only, no systems)
        film_data[name] = (FilmDeviceData(info, device.meshes[name], dtype, store_Q) if mine
                           else FilmDeviceData(info, device.meshes[name], dtype, False, geometry_only=True))
    for name, info in film_info_dict.items():
        mesh = device.meshes[name]
        if owned is not None and name not in owned:
            hole_systems[name] = {}
            continue
        fd = film_data[name]
        dev = fd.device
        inhomogeneous = info.lambda_info.inhomogeneous
        grad_Lambda_term = 0.0
        if inhomogeneous:
            # A = Q w - Lambda[cols] Del2 - grad_Lambda_term with (solve_film.py:181-185)
            #   grad_Lambda_term[j, k] = sum_d (grad_d @ Lambda)[j] grad_d[j, k]   (sparse, like grad).
            # Both sparse pieces go to the assembly kernel as ONE CSR in the Laplacian's slot, with
            # a unit Lambda:  A = Q w - 1 * (Del2 scaled per column + grad_Lambda_term).
            import scipy.sparse as sp

            ops = mesh.operators
            Lam = info.lambda_info.Lambda[:, 0].astype(np.float64)
            gx, gy = ops.gradient_x.tocsr(), ops.gradient_y.tocsr()
            grad_Lambda_term = (sp.diags(gx @ Lam) @ gx + sp.diags(gy @ Lam) @ gy).tocsr()
            corr = (ops.laplacian.tocsr().multiply(Lam[np.newaxis, :]) + grad_Lambda_term).tocsr()
            corr.sort_indices()
            fd.lap = tuple(_h2d(a, dev)
                           for a in (corr.indptr.astype(np.int64), corr.indices.astype(np.int64),
                                     corr.data.astype(np.float64)))
            fd.Lambda = torch.ones(fd.n, dtype=torch.float64, device=dev)

        def assemble(rows, cols, sign, fd=fd):
            return kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, rows, cols,
                                           sign=sign, dtype=dtype)

        hole_systems[name] = {}
        for hole_name, indices in info.hole_indices.items():
            ix_d = _h2d(indices.astype(np.int64), dev)
            if len(indices) == 0:  # a hole that contains no mesh vertex: an empty system, nothing to assemble
                hole_systems[name][hole_name] = LinearSystem(
                    indices=indices, indices_device=ix_d, grad_Lambda_term=grad_Lambda_term,
                    _assemble=lambda n=fd.n: np.zeros((n, 0), dtype=dtype))
                continue
            A_h = assemble(None, ix_d, 1.0)  # [n, ld]
            hole_systems[name][hole_name] = LinearSystem(
                indices=indices, A_device=A_h, indices_device=ix_d, grad_Lambda_term=grad_Lambda_term,
                _assemble=lambda A_h=A_h, k=len(indices): A_h[:, :k].cpu().numpy(),
            )
        targets = [("film", info.interior_indices)]
        if name in device.terminals:  # solve_film.py:220-263
            bix = np.asarray(info.boundary_indices, dtype=np.int64)
            bix_d = _h2d(bix, dev)
            A_b = assemble(None, bix_d, 1.0)
            terminal_systems[name] = TerminalSystems(
                film=name,
                boundary=LinearSystem(indices=bix, A_device=A_b, indices_device=bix_d,
                                      grad_Lambda_term=grad_Lambda_term,
                                      _assemble=lambda A_b=A_b, k=len(bix): A_b[:, :k].cpu().numpy()),
                holes=hole_systems[name])
            if info.hole_indices:
                targets.append(("film_without_boundary", info.interior_indices))
        for role, interior in targets:
            if role == "film":
                if info.hole_indices:  # solve_film.py:269-272
                    interior = _index_setdiff(interior, np.concatenate(list(info.hole_indices.values())), fd.n)
                if name in device.terminals:
                    interior = _index_setdiff(interior, info.boundary_indices, fd.n)  # :273-274
            ix_d = _h2d(interior.astype(np.int64), dev)
            ni = len(interior)

            def lu_route(ix_d=ix_d, ni=ni, fd=fd, assemble=assemble, name=name, assemble_only=False, factors=None,
                         padded=False):
                if factors is None:
                    if padded:   # buffer for the no-interchange route (kernels.lu_factor_nopivot_batch)
                        npad = kernels.lu_padded_n(ni)
                        return kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix_d, ix_d,
                                                       sign=-1.0, dtype=dtype, ld=kernels.padded_ld(npad, dtype),
                                                       alloc_rows=npad), ni
                    minusA = assemble(ix_d, ix_d, -1.0)      # -A, written once, factored in place
                    if assemble_only:
                        return minusA, ni
                    factors = kernels.lu_factor(minusA, ni)  # solve_film.py:279
                if factors.info > 0:
                    logger.warning(f"LU of film {name!r}: exactly singular U[{factors.info - 1}, "
                                   f"{factors.info - 1}] (LAPACK info = {factors.info}).")
                return factors

            def host_A(ix_d=ix_d, ni=ni, assemble=assemble):
                return assemble(ix_d, ix_d, 1.0)[:, :ni].cpu().numpy()

            S = None
            if ni == 0:  # no unknowns (e.g. every film vertex lies on the mesh boundary): empty system
                pending.append(((name, role), interior, ix_d, 0, None, None, lambda: np.zeros((0, 0), dtype=dtype), fd,
                                grad_Lambda_term))
                continue
            if inhomogeneous and method in ("cholesky", "mixed"):
                raise ValueError(f"Film {name!r}: Lambda(x, y) makes diag(w) A non-symmetric; "
                                 "use method='auto' or 'lu'.")
            if method == "mixed":
                npad = kernels.chol_padded_n(ni)
                S = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix_d, ix_d, sign=1.0,
                                            dtype="float32", row_scale=fd.w, lower_only=True,
                                            ld=kernels.padded_ld(npad, "float32"), alloc_rows=npad)
            elif method in ("auto", "cholesky") and not inhomogeneous:
                # S = diag(w) A is symmetric positive definite for a homogeneous film: Cholesky,
                # (1/3) n^3 flops, no pivoting; gf = -S^-1 (w[ix] * h)   (see chol.hip)
                npad = kernels.chol_padded_n(ni)
                S = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix_d, ix_d, sign=1.0,
                                            dtype=dtype, row_scale=fd.w, lower_only=True,
                                            ld=kernels.padded_ld(npad, dtype), alloc_rows=npad)
            pending.append(((name, role), interior, ix_d, ni, S, lu_route, host_A, fd, grad_Lambda_term))
    # All films are factored in one interleaved schedule (ssa_chol_factor_batch): the MFMA
    # trailing updates of the films alternate on the stream, each film's panel chain hides behind
    # the other films' updates.
    with_S = [p for p in pending if p[4] is not None]
    chols = dict(zip((p[0] for p in with_S), kernels.chol_factor_batch([(p[4], p[3]) for p in with_S], solve_block)))
    # what the Cholesky systems need besides the factor is enqueued BEFORE the host waits for the pivot reports
    # (one device-to-host copy for all films): nothing is left to launch between the factorization and the solve
    neg_w = {p[0]: (-p[7].w_t[p[2]]).contiguous() for p in with_S}
    # the mesh rows that are not unknowns of a film system (the self field's all-pairs rows, _solve_film_device):
    # host set arithmetic + upload, done here under the running factorization instead of inside the first pass
    exterior_d = {p[0]: _h2d(np.setdiff1d(np.arange(p[7].n, dtype=np.int64), p[1]), p[7].device)
                  for p in pending if p[3] > 0 and p[0][1] == "film"}
    kernels.fetch_chol_infos(list(chols.values()))
    # the films that take the LU route (method="lu", Lambda(x, y), a failed Cholesky) are factored together,
    # one stream per film: the panel chain of one film runs beside the trailing updates of the others
    lu_keys = [p[0] for p in pending if p[3] > 0 and (chols.get(p[0]) is None or chols[p[0]].info != 0)]
    if method in ("cholesky", "mixed") and any(chols.get(k) is not None for k in lu_keys):
        bad = [k[0] for k in lu_keys if chols.get(k) is not None]
        raise RuntimeError(f"diag(w) A of film {bad[0]!r} is not positive definite.")
    # First the no-interchange route (look-ahead, one schedule for all films): it applies whenever LAPACK's
    # partial pivoting would not swap rows, which is verified on the result; a film that fails the check is
    # assembled again and goes through the pivoting route (one stream per film).
    lu_batch = {}
    if lu_keys:
        routes = {p[0]: p[5] for p in pending}
        fast = kernels.lu_factor_nopivot_batch([routes[k](padded=True) for k in lu_keys])
        lu_batch = {k: f for k, f in zip(lu_keys, fast) if f is not None}
        redo = [k for k in lu_keys if k not in lu_batch]
        del fast
        if redo:
            logger.info(f"LU of {[k[0] for k in redo]}: row interchanges needed, factoring with partial pivoting.")
            lu_batch.update(zip(redo, kernels.lu_factor_batch([routes[k](assemble_only=True) for k in redo])))
    for key, interior, ix_d, ni, S, lu_route, host_A, fd, grad_Lambda_term in pending:
        name, role = key
        system = None
        chol = chols.get(key)
        if chol is not None:
            if chol.info == 0:
                system = LinearSystem(indices=interior, chol=chol, indices_device=ix_d,
                                      grad_Lambda_term=grad_Lambda_term,
                                      neg_w_device=neg_w[key], exterior_device=exterior_d.get(key),
                                      _lu_factorize=lu_route, _assemble=host_A)
            else:
                if method == "cholesky":
                    raise RuntimeError(f"diag(w) A of film {name!r} is not positive definite.")
                logger.warning(f"Film {name!r}: Cholesky pivot not positive, falling back to LU.")
                del chols[key]
                del S, chol
        if system is None and ni == 0:
            system = LinearSystem(indices=interior, indices_device=ix_d, grad_Lambda_term=grad_Lambda_term,
                                  _assemble=host_A)
        if system is None:
            factors = lu_route(factors=lu_batch[key])
            system = LinearSystem(indices=interior, factors=factors, indices_device=ix_d,
                                  grad_Lambda_term=grad_Lambda_term,
                                  rhs_indices_device=ix_d[factors.perm].contiguous(), _assemble=host_A,
                                  exterior_device=exterior_d.get(key))
        if role == "film":
            film_systems[name] = system
            if name in terminal_systems:
                terminal_systems[name].film_without_boundary_or_holes = system
                if terminal_systems[name].film_without_boundary is None and not film_info_dict[name].hole_indices:
                    terminal_systems[name].film_without_boundary = system
        else:
            terminal_systems[name].film_without_boundary = system
    return film_systems, hole_systems, terminal_systems, film_data


@dataclass
class FactorizedModel:
    """A pre-factorized model (``solver/solve.py:76-100``), reusable for any number of
    ``solve(model=...)`` calls; ``solve`` never mutates it."""

    device: Device
    film_info: Dict[str, FilmInfo]
    film_systems: Dict[str, LinearSystem]
    hole_systems: Dict[str, Dict[str, LinearSystem]]
    terminal_systems: Dict[str, object]
    terminal_currents: Dict[str, Dict[str, float]]
    circulating_currents: Dict[str, float]
    vortices: Sequence[Vortex]
    current_units: str
    film_data: Dict[str, FilmDeviceData] = field(default_factory=dict, repr=False)
    self_field_mode: str = "auto"
    method: str = "auto"

    def to_hdf5(self, h5group) -> None:
        """Saves what defines the model (``solver/solve.py:102-132``): device with meshes, currents,
        vortices, units and the factorization options.  The reference also writes every dense system
        and its LU factors; here the factors live in HBM (3-14 GB per film) and
        :meth:`from_hdf5` rebuilds them, which takes less time on an MI355X (0.1-0.5 s) than reading
        them back from storage would."""
        h5group.attrs["current_units"] = self.current_units
        h5group.attrs["self_field"] = self.self_field_mode
        h5group.attrs["method"] = self.method
        self.device.to_hdf5(h5group.create_group("device"))
        term_grp = h5group.create_group("terminal_currents")
        for film, terminals in self.terminal_currents.items():
            term_grp.create_group(film).attrs.update(terminals)
        h5group.create_group("circulating_currents").attrs.update(self.circulating_currents)
        vortex_grp = h5group.create_group("vortices")
        vortices = [v for info in self.film_info.values() for v in info.vortices]
        for i, vortex in enumerate(vortices):
            vortex.to_hdf5(vortex_grp.create_group(str(i)))

    @staticmethod
    def from_hdf5(h5group) -> "FactorizedModel":
        """``solver/solve.py:134-180``; re-factorizes on the GPU (see :meth:`to_hdf5`)."""
        vortex_grp = h5group["vortices"]
        return factorize_model(
            device=Device.from_hdf5(h5group["device"]), current_units=h5group.attrs["current_units"],
            terminal_currents={film: dict(grp.attrs) for film, grp in h5group["terminal_currents"].items()} or None,
            circulating_currents=dict(h5group["circulating_currents"].attrs),
            vortices=[Vortex.from_hdf5(vortex_grp[i]) for i in sorted(vortex_grp, key=int)],
            self_field=h5group.attrs.get("self_field", "auto"), method=h5group.attrs.get("method", "auto"))

    def set_circulating_currents(self, circulating_currents: Dict[str, Union[float, str]]) -> None:
        """``solver/solve.py:182-202``: no re-factorization needed."""
        currents = {k: current_to_float(v, self.current_units) for k, v in circulating_currents.items()}
        holes = self.device.holes
        for hole_name in currents:
            if hole_name not in holes:
                raise KeyError(f"Unknown hole {hole_name!r}.")
        self.circulating_currents = currents
        for info in self.film_info.values():
            info.circulating_currents = {h: c for h, c in currents.items() if h in info.hole_indices}

    def set_vortices(self, vortices: Sequence[Vortex]) -> None:
        """``solver/solve.py:204-217`` (the model's ``vortices`` becomes ``{film: tuple}`` there too)."""
        for info in self.film_info.values():
            info.vortices = []
        for vortex in vortices:
            self.film_info[vortex.film].vortices.append(vortex)
        self.vortices = {}
        for name, info in self.film_info.items():
            info.vortices = tuple(info.vortices)
            self.vortices[name] = info.vortices

    def vortex_column(self, film: str, j_film: int):
        """Column ``j_film`` of ``K = inv(A)`` of a film (``solve_film.py:541-544`` builds the whole
        inverse; one extra right-hand side through the existing factorization is enough), cached."""
        import torch

        from . import kernels

        cache = self.__dict__.setdefault("_vortex_columns", {})
        key = (film, int(j_film))
        if key not in cache:
            system = self.film_systems[film]
            fd = self.film_data[film]
            ni = len(system.indices)
            e = torch.zeros(ni, dtype=fd.tdtype, device=fd.device)
            if system.chol is not None and system.chol.dtype != fd.tdtype:
                raise NotImplementedError("method='mixed' does not handle trapped vortices.")
            if system.chol is not None:
                # A = diag(1/w) S  ->  inv(A) e_j = w_j inv(S) e_j ;  neg_w holds -w[ix]
                e[j_film] = -system.neg_w_device[j_film]
                cache[key] = kernels.chol_solve(system.chol, e)
            else:
                # lu_piv factors -A:  inv(A) e_j = -lu_solve(lu_piv, e_j)
                e[j_film] = 1.0
                cache[key] = -kernels.lu_solve(system.factors, e)
        return cache[key]

    def copy(self) -> "FactorizedModel":
        """Shallow copy (``solver/solve.py:219-220``)."""
        return _copy.copy(self)


# passes up to which a factorization is cheaper overall with 2048-row solve blocks (config H: the factorization saves
# 3-4 ms, a pass costs 0.12 ms more: break-even near 30)
FEW_PASSES = int(os.environ.get("SSA_FEW_PASSES", "24"))   # (0: never; an A/B aid)


def factorize_model(*, device: Device, current_units: str,
                    terminal_currents: Optional[Dict[str, Dict[str, Union[float, str]]]] = None,
                    circulating_currents: Optional[Dict[str, Union[float, str]]] = None,
                    vortices: Optional[Sequence[Vortex]] = None,
                    self_field: str = "auto", method: str = "auto",
                    placement: Optional[object] = None, expected_passes: Optional[int] = None) -> FactorizedModel:
    """``factorize_model`` (``solver/solve.py:223-287``).

    ``expected_passes`` (extension): how many ``solve_film`` passes per film the model is going to serve, if the
    caller knows (``solve(device=...)`` does: ``iterations + 1``).  Up to :data:`FEW_PASSES` the Cholesky factorization
    prepares its triangular solves on 2048-row diagonal blocks instead of 4096-row ones -- a quarter of the
    block-inverse flops (config H: 3-4 ms less factorization) for twice the dependent launches per solve (+ 0.12 ms
    per pass): ``include/superscreen_hip.h``, ``ssa_chol_factor_batch_blk``.  ``None``: a model for reuse, 4096.

    ``self_field`` (extension): how ``Q @ (w * g)`` (``solve_film.py:565``) is evaluated.
    ``"matrix_free"`` regenerates q_ij on the fly (2.2x faster than streaming a stored Q at n = 50k,
    and n^2 words less HBM); ``"dense"`` stores Q in the solve dtype like the reference and uses a
    GEMV; ``"london"`` uses, on the rows that are unknowns of the film system, the solved system itself
    -- the London equation ``H_applied + H_other + H_self = Laplacian(Lambda g)`` -- i.e. one sparse
    product, and the all-pairs sum only on the remaining rows (boundary, vacuum buffer, holes: 20-30 %
    of a mesh); it agrees with the all-pairs value to the residual of the solve (2e-12 relative at
    25k vertices per film, float64).  ``"auto"`` (default) = ``"london"`` for float64 films with a
    uniform Lambda and no vortices or terminals, ``"matrix_free"`` otherwise.
    ``method`` (extension): ``"auto"`` (default) / ``"cholesky"`` / ``"lu"`` / ``"mixed"`` (float32 factor, float64
    answers by iterative refinement: half the factorization time of a float64 device), see
    :func:`factorize_linear_systems`.
    ``placement`` (extension): a :class:`superscreen_amd.parallel.FilmPlacement`; this rank then
    assembles and factors only the films it owns (pass the same object to :func:`solve`).
    """
    if self_field not in ("auto", "london", "matrix_free", "dense"):
        raise ValueError(f"Unknown self_field mode {self_field!r}.")
    circulating_currents = {k: current_to_float(v, current_units)
                            for k, v in (circulating_currents or {}).items()}
    terminal_currents = {film: {k: current_to_float(v, current_units) for k, v in cur.items()}
                         for film, cur in (terminal_currents or {}).items()}
    for film_name, currents in terminal_currents.items():
        if sum(currents.values()):
            raise ValueError(f"Terminal currents in film {film_name!r} are not conserved.")
    vortices = list(vortices or [])
    if not device.meshes:
        raise ValueError("The device does not have a mesh. Call device.make_mesh() to generate it.")
    # the films' kernel-matrix row sums go out first: they only need the mesh, and run on the GPU while the host
    # works out the index sets (make_film_info: polygon tests, set arithmetic)
    if device.meshes:
        from . import _hip

        _hip.require_gpu()   # (no CPU fallback: HipLibraryError without the library or a GPU)
        mine_first = list(device.films) if placement is None else placement.mine(list(device.films))
        for name in mine_first:
            FilmDeviceData.start_row_sums(device.meshes[name], device.solve_dtype, self_field == "dense")
    try:
        film_info = make_film_info(device=device, vortices=vortices,
                                   circulating_currents=circulating_currents,
                                   terminal_currents=terminal_currents)
        owned = None if placement is None else placement.mine(list(device.films))
        # (method="mixed" runs 1 + MIXED_REFINEMENT_SWEEPS triangular solves per pass)
        solves = None if expected_passes is None else expected_passes * (1 + MIXED_REFINEMENT_SWEEPS if method == "mixed" else 1)
        solve_block = 2048 if (solves is not None and solves <= FEW_PASSES) else 4096
        film_systems, hole_systems, terminal_systems, film_data = factorize_linear_systems(
            device, film_info, store_Q=(self_field == "dense"), method=method, owned=owned, solve_block=solve_block)
    finally:
        for name in mine_first:
            FilmDeviceData.drop_row_sums(device.meshes[name], device.solve_dtype)
    model = FactorizedModel(device, film_info, film_systems, hole_systems, terminal_systems,
                            terminal_currents, circulating_currents, vortices, current_units,
                            film_data=film_data, self_field_mode=self_field, method=method)
    model.__dict__["_placement"] = placement
    return model


# ---------------------------------------------------------------------------------------
# solve_film / solve
# ---------------------------------------------------------------------------------------
@dataclass
class _DeviceFilmResult:
    g: object          # [n] solve dtype
    J: object          # [n, 2] float64
    self_field: object  # [n] solve dtype, raw (not yet divided by field_conversion)
    # rows of ``self_field`` that are still missing (zero): the all-pairs sum over the rows that are not unknowns,
    # left to ONE multi-vector launch for all iterates at the end of ``solve`` (None: nothing is missing)
    deferred_rows: object = None


def _system_solve(system: LinearSystem, h):
    """``lu_solve(system.lu_piv, h)`` for a right-hand side in natural order (device vector):
    through the Cholesky factor of ``diag(w) A`` if the system has one, else through the LU."""
    from . import kernels

    if system.chol is not None:
        return kernels.chol_solve(system.chol, kernels.row_scale(h, system.neg_w_device))
    return kernels.lu_solve(system.factors, h)


def stream_from_current_density(points: np.ndarray, J: np.ndarray) -> np.ndarray:
    """``solver/utils.py:440-463``: ``g(r) = g(r0) + int (z x J) . dl`` along ``points``."""
    from scipy import integrate

    zhat_cross_J = J[:, [1, 0]].copy()
    zhat_cross_J[:, 0] *= -1
    dl = np.diff(points, axis=0)
    integrand = np.sum(zhat_cross_J * dl, axis=1)
    return integrate.cumulative_trapezoid(integrand, initial=0)


def stream_from_terminal_current(points: np.ndarray, current: float) -> np.ndarray:
    """``solver/utils.py:466-488``: stream function along a terminal that sources ``current``
    uniformly and perpendicular to itself."""
    edge_lengths, unit_normals = path_vectors(points)
    J = current * unit_normals / np.sum(edge_lengths)
    g = stream_from_current_density(points, J)
    return g * current / g[-1]


def _terminal_transport(model: "FactorizedModel", name: str):
    """Device vectors ``(g_transport, Ha_transport)`` of a film with terminals
    (``solve_for_terminal_current_stream``, ``solve_film.py:308-390``, and the boundary effective
    field, ``:507-524``); they depend on the terminal currents only, so they are computed once per
    model and reused by every pass.  Boundary bookkeeping is host work on O(sqrt n) vertices; the
    gemvs, the two triangular solves and the all-pairs boundary field run on the GPU."""
    import torch

    from . import kernels

    cache = model.__dict__.setdefault("_transport", {})
    info = model.film_info[name]
    currents = dict(info.terminal_currents or {})
    key = (name, tuple(sorted(currents.items())))
    if key in cache:
        return cache[key]
    fd = model.film_data[name]
    device = model.device
    mesh = device.meshes[name]
    points = mesh.sites
    weights = mesh.operators.weights
    n = len(points)
    ts = model.terminal_systems[name]
    zeros = torch.zeros(n, dtype=fd.tdtype, device=fd.device)
    if not any(currents.values()):
        cache[key] = (zeros, zeros.clone())
        return cache[key]

    def put(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(fd.device).to(fd.tdtype)

    def minus_A_times(system: LinearSystem, g_d, out):
        kernels.gemv(system.A_device, n, len(system.indices), g_d, xidx=system.indices_device, y=out,
                     alpha=-1.0, beta=1.0)

    boundary_indices = ts.boundary.indices
    boundary_points = points[boundary_indices]
    # 1. stream function on the boundary (:342-353)
    g = np.zeros(n)
    for terminal in device.terminals[name]:
        current = currents[terminal.name]
        ix_boundary = np.sort(terminal.contains_points(boundary_points, index=True))
        remaining_boundary = boundary_indices[ix_boundary[-1]:]
        ix_terminal = boundary_indices[ix_boundary]
        stream = stream_from_terminal_current(points[ix_terminal], -current)
        g[ix_terminal[:-1]] += stream
        g[remaining_boundary] += stream[-1]
    g = g - np.max(g) + np.ptp(g) / 2
    ha = torch.zeros(n, dtype=fd.tdtype, device=fd.device)
    minus_A_times(ts.boundary, put(g), ha)
    # 2. interior, holes ignored (:358-364)
    fwb = ts.film_without_boundary
    h = kernels.film_rhs(zeros, None, ha, fwb.indices_device)          # = -Ha_eff[indices]
    g[fwb.indices] = _system_solve(fwb, h).cpu().numpy()
    if ts.holes:
        # holes: weighted average of the hole-free answer, then re-solve (:366-385)
        ha = torch.zeros(n, dtype=fd.tdtype, device=fd.device)
        for system in ts.holes.values():
            ix = system.indices
            g[ix] = np.average(g[ix], weights=weights[ix])
        g_d = put(g)
        for system in ts.holes.values():
            minus_A_times(system, g_d, ha)
        minus_A_times(ts.boundary, g_d, ha)
        sys3 = ts.film_without_boundary_or_holes
        h = kernels.film_rhs(zeros, None, ha, sys3.indices_device)
        g[sys3.indices] = _system_solve(sys3, h).cpu().numpy()
    g_transport = put(g)
    # effective field of the boundary stream (:510-524; numba kernel _get_boundary_effective_field
    # :393-412): sum_j stream_j (dr . -n_j) len_j / (4 pi |dr|^3) over the boundary edges -- the
    # z component of a "sheet" made of the edge centres with J = (b, -a), (a, b) = -stream len n
    b_idx = info.boundary_indices
    boundary_sites = points[b_idx]
    boundary_stream = g[b_idx]
    centers = 0.5 * (boundary_sites + np.roll(boundary_sites, -1, axis=0))
    boundary_stream = 0.5 * (boundary_stream + np.roll(boundary_stream, -1, axis=0))
    lengths, normals = path_vectors(close_curve(boundary_sites))
    ab = -(boundary_stream * lengths)[:, None] * normals
    ev = np.column_stack([points, np.zeros(n)])
    f64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(fd.device)  # noqa: E731
    ha_t = kernels.sheet_field(f64(centers), f64(np.ones(len(centers))),
                               f64(np.column_stack([ab[:, 1], -ab[:, 0]])), 0.0, f64(ev),
                               1.0 / (4 * np.pi), False).to(fd.tdtype)
    cache[key] = (g_transport, ha_t)
    return cache[key]


def _solve_film_device(model: FactorizedModel, name: str, applied_d, other_d,
                       check_inversion: bool, vortex_flux_value: float = 0.0,
                       defer_exterior: bool = False) -> _DeviceFilmResult:
    """Device part of ``solve_film`` (``solver/solve_film.py:486-565``) for one film: ``_solve_film_steps`` with its
    Cholesky solve(s) done on the spot."""
    return _solve_films_device(model, [name], {name: applied_d}, None if other_d is None else {name: other_d},
                               check_inversion, vortex_flux_value, defer_exterior)[name]


def _solve_films_device(model: FactorizedModel, names: Sequence[str], applied_d, other_d, check_inversion: bool,
                        vortex_flux_value: float = 0.0, defer_exterior: bool = False,
                        pass_cache: Optional[dict] = None) -> Dict[str, _DeviceFilmResult]:
    """``_solve_film_device`` for several films of one pass (independent of each other, ``solver/solve.py:517-536``):
    the Cholesky solves of all of them go out together (``kernels.chol_solve_batch``: the block steps of the
    triangular solves side by side in one launch each).  A film asks for one solve (several with ``method="mixed"``:
    one per refinement sweep); the films that still wait for one are served together, round after round."""
    from . import kernels

    results, waiting = {}, []
    for name in names:
        steps = _solve_film_steps(model, name, applied_d[name], None if other_d is None else other_d[name],
                                  check_inversion, vortex_flux_value, defer_exterior, pass_cache)
        try:
            waiting.append((name, steps) + tuple(next(steps)))
        except StopIteration as stop:
            results[name] = stop.value
    while waiting:
        by_dtype: Dict[object, list] = {}
        for item in waiting:
            by_dtype.setdefault(item[2].dtype, []).append(item)
        waiting = []
        for group in by_dtype.values():
            solved = kernels.chol_solve_batch([item[2] for item in group], [item[3] for item in group], padded=True)
            for (name, steps, _, _), gf in zip(group, solved):
                try:
                    waiting.append((name, steps) + tuple(steps.send(gf)))
                except StopIteration as stop:
                    results[name] = stop.value
    return {name: results[name] for name in names}


def _solve_film_steps(model: FactorizedModel, name: str, applied_d, other_d,
                      check_inversion: bool, vortex_flux_value: float = 0.0,
                      defer_exterior: bool = False, pass_cache: Optional[dict] = None):
    """Device part of ``solve_film`` (``solver/solve_film.py:486-565``) as a generator: it yields
    ``(CholFactors, right-hand side padded to chol_padded_n)`` where the film's system goes through its Cholesky factor, is sent the solution
    and returns the ``_DeviceFilmResult`` (films on the LU route never yield).  ``defer_exterior``: where the self field
    of the film interior comes from the London equation, leave its all-pairs part (the rows that are not unknowns)
    to the caller, which evaluates it for all iterates at once (``_DeviceFilmResult.deferred_rows``)."""
    import torch

    from . import kernels

    fd = model.film_data[name]
    info = model.film_info[name]
    system = model.film_systems[name]
    has_terminals = name in model.device.terminals
    fixed = None if pass_cache is None else pass_cache.get(name)
    if fixed is None:
        g = torch.zeros(fd.n, dtype=fd.tdtype, device=fd.device)
        ha_eff = torch.zeros(fd.n, dtype=fd.tdtype, device=fd.device)
        for hole_name, hs in model.hole_systems[name].items():
            current = info.circulating_currents.get(hole_name, 0)
            if len(hs.indices) == 0:
                continue
            kernels.index_add_scalar(g, hs.indices_device, current)        # g[hole] += I_circ
            kernels.gemv(hs.A_device, fd.n, len(hs.indices), g, xidx=hs.indices_device,
                         y=ha_eff, alpha=-1.0, beta=1.0)                      # Ha_eff += -(A @ g[ix])
        if has_terminals:  # solve_film.py:505-524
            g_transport, ha_transport = _terminal_transport(model, name)
            g += g_transport
            ha_eff += ha_transport
        if pass_cache is not None:
            # the holes' and terminals' part of g and of the effective field depends neither on the applied field
            # nor on the iteration (solve_film.py:498-524 recomputes it in every call: an [n, n_hole] product per hole):
            # evaluated in the first pass of a solve, copied in the others
            pass_cache[name] = (g.clone(), ha_eff)
    else:
        g, ha_eff = fixed[0].clone(), fixed[1]     # (ha_eff is only read below)
    if len(system.indices) == 0:
        gf = None
    elif system.chol is not None:
        h_nat = kernels.film_rhs(applied_d, other_d, ha_eff, system.indices_device)
        # the right-hand side in a buffer padded as the factorization is (zero tail), kept for all passes of a solve:
        # the triangular solves then run where it is, without staging copies
        rhs = None if pass_cache is None else pass_cache.get(("rhs", name))
        if rhs is None:
            rhs = torch.zeros(kernels.chol_padded_n(system.chol.n), dtype=fd.tdtype, device=fd.device)
            if pass_cache is not None:
                pass_cache[("rhs", name)] = rhs
        kernels.row_scale(h_nat, system.neg_w_device, out=rhs[:system.chol.n])
        if system.chol.dtype != fd.tdtype:   # method="mixed": float32 factor, float64 answer
            if info.vortices:
                raise NotImplementedError("method='mixed' does not handle trapped vortices.")
            gf = yield from _mixed_solve(fd, system, rhs, pass_cache, name)
        else:
            gf = yield system.chol, rhs
    else:
        h = kernels.film_rhs(applied_d, other_d, ha_eff, system.rhs_indices_device)
        if check_inversion:
            h_nat = kernels.film_rhs(applied_d, other_d, ha_eff, system.indices_device)
        gf = kernels.lu_solve_permuted(system.factors, h)                # = lu_solve(lu_piv, h)
    if check_inversion and gf is not None:  # solve_film.py:533-540: warn, never raise
        A = torch.from_numpy(system.A).to(fd.device)
        hsim = -(kernels.gemv(A, len(system.indices), len(system.indices), gf))
        if not np.allclose(hsim.cpu().numpy(), h_nat.cpu().numpy()):
            err = (hsim - h_nat).abs().max().item()
            logger.warning(f"Unable to solve for stream function in {name!r}), maximum error {err:.3e}.")
    if gf is not None:
        kernels.scatter_add(g, system.indices_device, gf)
    if info.vortices:  # solve_film.py:541-554, Eq. 28 in [Brandt]
        mesh = model.device.meshes[name]
        points = mesh.sites
        weights = mesh.operators.weights
        for vortex in info.vortices:
            xy = (vortex.x, vortex.y)
            j_film = int(np.argmin(np.linalg.norm(points[system.indices] - xy, axis=1)))
            j_device = int(np.argmin(np.linalg.norm(points - xy, axis=1)))
            scale = vortex_flux_value * vortex.nPhi0 / float(weights[j_device])
            g_vortex = kernels.scale(model.vortex_column(name, j_film), scale)
            kernels.scatter_add(g, system.indices_device, g_vortex)
    J = kernels.current_density(*fd.grad, g)
    if has_terminals:
        # solve_film.py:557-562: Biot-Savart of the per-triangle currents (numba kernel
        # _biot_savart_within_film, :415-437) = z component of a sheet made of the triangle centroids
        tri = fd.triangle_data(model.device.meshes[name])
        J_tri = kernels.current_density(*tri["grad"], g)
        sf = kernels.sheet_field(tri["centroids"], tri["areas"], J_tri, 0.0, tri["sites3"],
                                 1.0 / (4 * np.pi), False).to(fd.tdtype)
    elif model.self_field_mode == "dense":
        sf = kernels.gemv(fd.Q, fd.n, fd.n, g, xscale=fd.w_t)
    elif (model.self_field_mode in ("auto", "london") and fd.tdtype == torch.float64 and not info.vortices
          and not info.lambda_info.inhomogeneous and len(system.indices) > 0):
        # interior rows: the solved system is the London equation, H_applied + H_other + H_self =
        # Laplacian(Lambda g), i.e. the self field costs one sparse product there; the all-pairs sum
        # Q @ (w g) is only needed on the rows that are not unknowns (boundary, buffer, holes)
        if system.exterior_device is None:
            exterior = np.setdiff1d(np.arange(fd.n, dtype=np.int64), system.indices)
            system.exterior_device = torch.from_numpy(exterior).to(fd.device)
        deferred = system.exterior_device if (defer_exterior and system.exterior_device.numel() > 0) else None
        sf = torch.zeros_like(g) if deferred is not None else torch.empty_like(g)
        kernels.london_field_rows(*fd.lap, fd.Lambda, g, applied_d, other_d, system.indices_device, sf)
        if deferred is None:
            kernels.self_field_rows(fd.xy, fd.w, fd.qdiag, g, system.exterior_device, sf)
        return _DeviceFilmResult(g=g, J=J, self_field=sf, deferred_rows=deferred)
    else:
        sf = kernels.self_field(fd.xy, fd.w, fd.qdiag, g)
    return _DeviceFilmResult(g=g, J=J, self_field=sf)


MIXED_REFINEMENT_SWEEPS = 2   # each sweep shrinks the error of the float32 solve by ~ 1e-4 (measured, config H)


def _mixed_solve(fd: "FilmDeviceData", system: LinearSystem, rhs, pass_cache: Optional[dict], name: str):
    """``S x = rhs`` with ``S = diag(w) A`` factored in FLOAT32 (``factorize_model(method="mixed")``) and the answer
    refined in float64: ``x <- x + S32^-1 (rhs - S x)``, the residual against the EXACT float64 ``S`` -- evaluated matrix
    free, as the assembly kernel would: ``(S x)_i = w_i ((Q (w x))_i - (Del2 (Lambda x))_i)`` on the unknowns' rows, the
    all-pairs sum (``ssa_self_field_rows``) and the sparse Laplacian product (``ssa_london_field_rows``).  A generator
    like :func:`_solve_film_steps`: yields ``(factor, padded float32 right-hand side)``, is sent the solution."""
    import torch

    from . import kernels

    n, npad = system.chol.n, rhs.numel()
    ix = system.indices_device
    keep = {} if pass_cache is None else pass_cache.setdefault(("mixed", name), {})
    if not keep:
        keep.update(rhs32=torch.zeros(npad, dtype=torch.float32, device=fd.device),
                    full=torch.zeros(fd.n, dtype=torch.float64, device=fd.device),
                    q=torch.empty(fd.n, dtype=torch.float64, device=fd.device),
                    lap=torch.empty(fd.n, dtype=torch.float64, device=fd.device),
                    zero=torch.zeros(fd.n, dtype=torch.float64, device=fd.device),
                    w_ix=(-system.neg_w_device).to(torch.float64))
    rhs32, full = keep["rhs32"], keep["full"]
    rhs32.copy_(rhs)
    x = (yield system.chol, rhs32).to(torch.float64)          # (a view of rhs32's first n elements -> a new tensor)
    for _ in range(MIXED_REFINEMENT_SWEEPS):
        full.zero_()
        full[ix] = x
        kernels.self_field_rows(fd.xy, fd.w, fd.qdiag, full, ix, keep["q"])
        kernels.london_field_rows(*fd.lap, fd.Lambda, full, keep["zero"], None, ix, keep["lap"])
        residual = rhs[:n] - keep["w_ix"] * (keep["q"][ix] - keep["lap"][ix])
        rhs32[:n] = residual
        x = x + (yield system.chol, rhs32).to(torch.float64)
    return x


_copy_streams: Dict[int, object] = {}


class _StagedPass:
    """Results of one pass on their way to the host: device-to-host copies run on a side stream
    into pinned buffers while the main stream already computes the next Jacobi iteration."""

    def __init__(self, results: Dict[str, _DeviceFilmResult], other_d, films: Sequence[str]):
        import torch

        dev = next(iter(results.values())).g.device
        # Three or more films: the copies go out on the compute stream itself, between two passes (4 x 5 n values:
        # 65 us per pass).  HIP maps the streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues per
        # priority level, and when the side stream lands on the compute stream's queue its copies and event waits
        # serialize with the passes: measured 112 instead of 91 ms for the 11 passes of a four-film stack at the
        # default queue count (two films: 22.6 against 22.0 ms -- they keep the side stream).
        inline = len(films) >= 3
        stream = torch.cuda.current_stream(dev) if inline else _copy_streams.get(dev.index)
        if stream is None:
            stream = _copy_streams[dev.index] = torch.cuda.Stream(device=dev)
        ready = torch.cuda.Event()
        ready.record()  # on the compute stream, after the pass
        self.host: Dict[str, Dict[str, object]] = {}
        self._keep = (results, other_d)  # device tensors stay alive until the copies are done
        with torch.cuda.stream(stream):
            if not inline:
                stream.wait_event(ready)
            for name in films:
                res = results[name]
                items = {"g": res.g, "J": res.J, "self_field": res.self_field}
                if other_d is not None:
                    items["other"] = other_d[name]
                out = {}
                for key, t in items.items():
                    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    if not inline:
                        t.record_stream(stream)
                    out[key] = h
                self.host[name] = out
            self.done = torch.cuda.Event()
            self.done.record(stream)

    def film_solution(self, name: str, applied_h: np.ndarray, conv: float) -> FilmSolution:
        """Host epilogue of ``solve_film`` (``solver/solve_film.py:566-574``)."""
        self.done.synchronize()
        self._keep = None
        h = self.host[name]
        other = h["other"].numpy() / conv if "other" in h else None
        return FilmSolution(
            stream=h["g"].numpy(),
            current_density=h["J"].numpy(),
            applied_field=applied_h / conv,
            self_field=h["self_field"].numpy() / conv,
            field_from_other_films=other,
        )


def solve(device: Optional[Device] = None, *, model: Optional[FactorizedModel] = None,
          applied_field: Optional[Callable] = None,
          terminal_currents: Optional[Dict[str, Dict[str, Union[float, str]]]] = None,
          circulating_currents: Optional[Dict[str, Union[float, str]]] = None,
          vortices: Optional[Sequence[Vortex]] = None, field_units: str = "mT",
          current_units: str = "uA", check_inversion: bool = False, iterations: int = 0,
          return_solutions: bool = True, save_path=None, log_level: Optional[int] = None,
          progress_bar: bool = True, tolerance: Optional[float] = None,
          coupling: Optional[object] = None, placement: Optional[object] = None,
          _solver: str = "superscreen_amd.solve") -> Optional[List[Solution]]:
    """``solve`` (``solver/solve.py:290-549``): same arguments, same Jacobi scheme, same list of
    ``iterations + 1`` Solutions (1 for a single film or ``iterations < 1``).

    Extensions: ``tolerance`` stops the fixed-count loop early once
    ``max_f max|g_k - g_{k-1}| / max|g_k| < tolerance`` (the reference has no convergence test,
    SURVEY.md quirk 6; ``iterations`` stays the upper bound); ``coupling`` is an optional
    :class:`superscreen_amd.parallel.CouplingPlan` that spreads the inter-film Biot-Savart sums
    over several GPUs (one RCCL all-reduce per iteration); ``placement`` an optional
    :class:`superscreen_amd.parallel.FilmPlacement` (owner-computes: this rank solves the films it
    factored and the complete coupling field of those films, the owners broadcast the O(n) result
    vectors after every pass, every rank returns the same Solutions).
    """
    import torch

    from . import kernels

    if log_level is not None:
        logging.basicConfig(level=log_level)

    if model is None:
        if device is None:
            raise ValueError("Either a model or a device must be provided.")
        # (the plain cold call of the reference: the factorization serves this solve only -- and says so)
        model = factorize_model(device=device, current_units=current_units,
                                terminal_currents=terminal_currents,
                                circulating_currents=circulating_currents, vortices=vortices,
                                expected_passes=(iterations + 1) if len(device.films) > 1 else 1)
    else:
        if (device is not None or terminal_currents is not None
                or circulating_currents is not None or vortices is not None):
            raise ValueError("If model argument is provided, device, terminal_currents,"
                             " circulating_currents, and vortices must be None.")
    if not isinstance(model, FactorizedModel):
        raise TypeError(f"model must be an instance of FactorizedModel (got {type(model)}).")

    device = model.device
    film_info = model.film_info
    current_units = model.current_units
    if not device.meshes:
        raise ValueError("The device does not have a mesh. Call device.make_mesh() to generate it.")
    dtype = device.solve_dtype
    meshes = device.meshes
    applied_field = applied_field or ConstantField(0)
    conv = field_conversion_factor(field_units, current_units, length_units=device.length_units)

    applied_h, applied_d = {}, {}
    for film, mesh in meshes.items():
        z0 = film_info[film].z0 * np.ones(len(mesh.sites))
        Hz = np.squeeze(np.asarray(applied_field(mesh.sites[:, 0], mesh.sites[:, 1], z0)) * conv)
        Hz = np.asarray(Hz * np.ones(len(mesh.sites)) if Hz.ndim == 0 else Hz).astype(dtype, copy=False)
        if Hz.ndim != 1:
            raise ValueError(f"Expected applied_field to return a 1D vector, got a {Hz.shape[1]}D vector.")
        applied_h[film] = Hz
        applied_d[film] = _h2d(Hz, model.film_data[film].device)
    vflux = vortex_flux(current_units, device.length_units)  # solve.py:441-442

    solution_kwargs = dict(applied_field_func=applied_field, field_units=field_units,
                           current_units=current_units,
                           circulating_currents=model.circulating_currents,
                           terminal_currents=model.terminal_currents, vortices=model.vortices,
                           solver=_solver)
    solutions: List[Solution] = []
    films = list(device.films)
    if placement is None:
        placement = model.__dict__.get("_placement")
    if placement is not None and coupling is not None:
        raise ValueError("Use either a CouplingPlan or a FilmPlacement, not both.")
    mine = films if placement is None else placement.mine(films)
    if placement is not None and any(model.film_systems.get(f) is None for f in mine):
        raise ValueError("The model was not factorized with this placement.")

    # with a placement every rank computes the same Solutions; only rank 0 writes them
    writes = save_path is not None and (placement is None or placement.rank == 0)
    keep = return_solutions or writes
    # The self field only goes into the returned Solutions (the iteration feeds on the sheet currents), and its
    # all-pairs part - the rows that are not unknowns, where the London equation does not give it - costs one
    # evaluation of r^-3 per pair whatever the number of vectors: it is evaluated for ALL iterates in one
    # multi-vector launch after the last pass (config H: 0.44 ms instead of 11 x 0.23 ms) and patched into the
    # Solutions before they are returned.  Not with a file (iterates are written as they come) and not with films
    # spread over ranks (the owners' vectors travel after every pass).
    # INVARIANT while ``batch_exterior`` is set: per pass, ``other_d[film]`` (the coupling field) and the self field are
    # valid on ``system.indices`` (the unknowns' rows) only -- all the iteration reads (``h = Hz[indices] - ...``); the
    # other rows are patched into the returned Solutions after the last pass.  ``solve(save_path=...)`` and placement
    # solves evaluate every row in every pass, so their fields agree with this route to rounding, not bit for bit
    # (tests/test_solve_gpu.py::test_output_only_rows_batched_equal_per_pass_route).
    batch_exterior = return_solutions and save_path is None and placement is None and len(films) >= 2 and iterations >= 1
    deferred: List[Dict[str, _DeviceFilmResult]] = []   # one entry per pass: the results with missing rows
    pass_cache: Dict[str, object] = {}                  # per film: what is the same in every pass of this solve
    coupling_sources: List[Dict[str, object]] = []      # one entry per iteration: the sheet currents it started from

    def run_pass(other_d):
        results = _solve_films_device(model, mine, applied_d, other_d, check_inversion, vflux,
                                      defer_exterior=batch_exterior, pass_cache=pass_cache)
        if batch_exterior:
            deferred.append({name: res for name, res in results.items() if res.deferred_rows is not None})
        if placement is not None:
            # owners broadcast their films' result vectors (and coupling fields): O(n) each
            fds = model.film_data
            payload = {f: ({"g": results[f].g, "J": results[f].J, "self_field": results[f].self_field}
                           if f in results else {}) for f in films}
            shapes = {f: {"g": (fds[f].n,), "J": (fds[f].n, 2), "self_field": (fds[f].n,)} for f in films}
            dtypes = {f: {"g": fds[f].tdtype, "J": torch.float64, "self_field": fds[f].tdtype} for f in films}
            if other_d is not None:
                for f in films:
                    if f in results:
                        payload[f]["other"] = other_d[f]
                    shapes[f]["other"], dtypes[f]["other"] = (fds[f].n,), fds[f].tdtype
            placement.share(films, payload, shapes, dtypes, fds[films[0]].device)
            for f in films:
                if f not in results:
                    results[f] = _DeviceFilmResult(g=payload[f]["g"], J=payload[f]["J"],
                                                   self_field=payload[f]["self_field"])
                if other_d is not None:
                    other_d[f] = payload[f]["other"]
        return results

    n_saved = [0]
    h5file = [None]

    def close_file():
        if h5file[0] is not None:
            h5file[0].close()
            h5file[0] = None

    def package(staged: _StagedPass):
        """Builds the iteration's Solution on the host; with ``save_path`` it also goes to the file:
        the device once in the root group, every iterate in a group named after its index
        (``solver/solve.py:474-483, 539-547``; read back by ``Solution.load_solutions``).  The file
        stays open for the whole solve and is written out once, when it is closed."""
        fs = {name: staged.film_solution(name, applied_h[name], conv) for name in films}
        solution = Solution(device=device, film_solutions=fs, **solution_kwargs)
        if writes:
            from . import io

            if h5file[0] is None:
                h5file[0] = io.open_file(save_path, "x")
                device.to_hdf5(h5file[0].create_group("device"))
            solution.to_hdf5(h5file[0].create_group(str(n_saved[0])), device_path="/device")
            n_saved[0] += 1
        if return_solutions:
            solutions.append(solution)

    results = run_pass(None)
    pending = _StagedPass(results, None, films) if keep else None
    if len(films) < 2 or iterations < 1:
        try:
            if keep:
                package(pending)
        finally:
            close_file()
        return solutions if return_solutions else None

    try:
        for it in range(iterations):
            other_d = {name: torch.zeros(model.film_data[name].n, dtype=model.film_data[name].tdtype,
                                         device=model.film_data[name].device) for name in films}
            if coupling is not None:
                coupling.accumulate(model, results, other_d)
            elif batch_exterior:
                # The iteration reads the coupling field only on the rows that are unknowns (h = Hz[indices] - ...,
                # solve_film.py:526-529); the other rows only go into the returned Solutions.  Per pass the
                # all-pairs sums run over the unknowns' rows (a compact copy of their coordinates); the remaining
                # rows of ALL passes are evaluated in one multi-vector launch per film pair after the last pass.
                coupling_sources.append({name: results[name].J for name in films})
                for tgt in films:
                    t = model.film_data[tgt]
                    rows = model.film_systems[tgt].indices_device
                    if rows is None or rows.numel() == 0:
                        continue
                    cache = model.film_systems[tgt].__dict__
                    xy_rows = cache.get("_xy_rows")
                    if xy_rows is None:
                        xy_rows = cache["_xy_rows"] = t.xy.index_select(0, rows).contiguous()
                    compact = torch.empty(rows.numel(), dtype=t.tdtype, device=t.device)
                    first = True
                    for src in films:
                        if src == tgt:
                            continue
                        s = model.film_data[src]
                        kernels.biot_savart(s.xy, s.w_t, results[src].J, xy_rows,
                                            film_info[tgt].z0 - film_info[src].z0, compact,
                                            accumulate=not first, src_begin=s.src_range[0], src_end=s.src_range[1])
                        first = False
                    other_d[tgt].index_copy_(0, rows, compact)
            else:
                # owner-computes: only this rank's target films; with helper ranks (more ranks than films) a source
                # slice of the film of this rank's group, summed inside the group before the owner solves
                targets = mine if placement is None else placement.coupling_targets(films)
                for src, tgt in itertools.product(films, repeat=2):  # solve.py:499-515
                    if src == tgt or tgt not in targets:
                        continue
                    s, t = model.film_data[src], model.film_data[tgt]
                    b, e = s.src_range if placement is None else placement.source_slice(*s.src_range)
                    if e > b:
                        kernels.biot_savart(s.xy, s.w_t, results[src].J, t.xy,
                                            film_info[tgt].z0 - film_info[src].z0, other_d[tgt],
                                            accumulate=True, src_begin=b, src_end=e)
                if placement is not None:
                    placement.reduce_coupling([other_d[tgt] for tgt in targets])
            prev = results
            results = run_pass(other_d)  # Jacobi: every film sees the previous iterate
            if keep:
                # the previous iterate is unpacked on the host while the GPU works on this one
                staged = _StagedPass(results, other_d, films)
                package(pending)
                pending = staged
            if tolerance is not None:
                change = max(((results[n].g - prev[n].g).abs().max() / results[n].g.abs().max()).item()
                             for n in films)
                logger.debug(f"iteration {it + 1}: relative change {change:.3e}")
                if change < tolerance:
                    break
        # (the multi-vector launches go out before the host unpacks the last iterate: they run meanwhile)
        patch = _enqueue_exterior_self_fields(model, deferred) if batch_exterior else None
        patch_c = _enqueue_exterior_coupling(model, coupling_sources) if (batch_exterior and coupling_sources) else None
        if keep:
            package(pending)
        if patch is not None:
            _patch_exterior_self_fields(model, patch, solutions, conv)
        if patch_c is not None:
            _patch_exterior_coupling(model, patch_c, solutions, conv)
    finally:
        close_file()
    return solutions if return_solutions else None


def _enqueue_exterior_self_fields(model: FactorizedModel, deferred):
    """The all-pairs part of the self field (``Q @ (w * g)`` on the rows that are not unknowns,
    ``solver/solve_film.py:565``) of every iterate in ``deferred`` in one multi-vector launch per film, with the
    device-to-host copies of the results behind them (pass k = ``deferred[k]``).  Returns what
    ``_patch_exterior_self_fields`` needs."""
    import torch

    from . import kernels

    names = sorted({name for per_pass in deferred for name in per_pass})
    staged = []
    for name in names:
        passes = [k for k, per_pass in enumerate(deferred) if name in per_pass]
        fd = model.film_data[name]
        rows = deferred[passes[0]][name].deferred_rows
        G = torch.stack([deferred[k][name].g for k in passes], dim=1).contiguous()     # [n, passes]
        out = torch.empty_like(G)
        kernels.self_field_multi_rows(fd.xy, fd.w, fd.qdiag, G, rows, out)
        vals = out.index_select(0, rows)                                                  # [rows, passes]
        host = torch.empty(vals.shape, dtype=vals.dtype, pin_memory=True)
        host.copy_(vals, non_blocking=True)
        staged.append((name, passes, rows.numel(), host))
    done = torch.cuda.Event()
    done.record()
    return staged, done


def _exterior_rows_host(model: FactorizedModel, name: str) -> np.ndarray:
    """The mesh rows of film ``name`` that are not unknowns of its system (mesh-only data: kept with the system)."""
    system = model.film_systems[name]
    ext = system.__dict__.get("_exterior_host")
    if ext is None:
        mask = np.ones(model.film_data[name].n, dtype=bool)
        mask[system.indices] = False
        ext = system.__dict__["_exterior_host"] = np.flatnonzero(mask)
    return ext


def _enqueue_exterior_coupling(model: FactorizedModel, coupling_sources):
    """The field from the other films (``solver/solve.py:499-515``) on the rows that are NOT unknowns, for every
    iteration in ``coupling_sources`` (iteration k started from the sheet currents ``coupling_sources[k]``) in one
    multi-vector launch per ordered film pair, with the device-to-host copies behind them."""
    import torch

    from . import kernels

    films = list(coupling_sources[0])
    nvec = len(coupling_sources)
    staged = []
    for tgt in films:
        t = model.film_data[tgt]
        ext = _exterior_rows_host(model, tgt)
        if len(ext) == 0:
            continue
        system = model.film_systems[tgt]
        rows = system.exterior_device
        if rows is None:
            rows = system.exterior_device = torch.from_numpy(ext).to(t.device)
        out = torch.zeros((t.n, nvec), dtype=t.tdtype, device=t.device)
        for src in films:
            if src == tgt:
                continue
            s = model.film_data[src]
            b, e = s.src_range
            J = torch.stack([coupling_sources[k][src][b:e] for k in range(nvec)], dim=1).contiguous()   # [ns, nvec, 2]
            kernels.biot_savart_multi(s.xy[b:e], s.w_t[b:e], J, t.xy, model.film_info[tgt].z0 - model.film_info[src].z0,
                                      out, accumulate=True, rows=rows)
        vals = out.index_select(0, rows)
        host = torch.empty(vals.shape, dtype=vals.dtype, pin_memory=True)
        host.copy_(vals, non_blocking=True)
        staged.append((tgt, ext, host))
    done = torch.cuda.Event()
    done.record()
    return staged, done


def _patch_exterior_coupling(model: FactorizedModel, patch, solutions: List[Solution], conv: float) -> None:
    """Writes the rows computed by ``_enqueue_exterior_coupling`` into
    ``solutions[k + 1].film_solutions[name].field_from_other_films`` (iteration k produced Solution k + 1)."""
    staged, done = patch
    done.synchronize()
    for name, ext, host in staged:
        vals = host.numpy() / conv
        for k in range(vals.shape[1]):
            if k + 1 < len(solutions):
                solutions[k + 1].film_solutions[name].field_from_other_films[ext] = vals[:, k]


def _patch_exterior_self_fields(model: FactorizedModel, patch, solutions: List[Solution], conv: float) -> None:
    """Writes the rows computed by ``_enqueue_exterior_self_fields`` into
    ``solutions[k].film_solutions[name].self_field`` (pass k = ``solutions[k]``)."""
    staged, done = patch
    rows_h = {name: _exterior_rows_host(model, name) for name, _, _, _ in staged}
    done.synchronize()
    for name, passes, _, host in staged:
        vals = host.numpy() / conv
        for col, k in enumerate(passes):
            solutions[k].film_solutions[name].self_field[rows_h[name]] = vals[:, col]
