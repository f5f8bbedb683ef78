"""Thin, typed wrappers around the C ABI (``include/superscreen_hip.h``) operating on torch
CUDA tensors.  One function per entry point; the docstrings name the reference call site
each one replaces (paths relative to ``/root/reference/superscreen/``).

PyTorch provides device buffers and the stream; all arithmetic happens in the HIP library.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _hip
from ._hip import check, current_stream, dtype_code, load_library, ptr


def _tdtype(dtype) -> torch.dtype:
    return torch.float64 if dtype_code(dtype) == _hip.SSA_F64 else torch.float32


def padded_ld(n: int, dtype) -> int:
    """Leading dimension that keeps every row 128-byte aligned (full cache lines)."""
    per_line = 128 // (8 if dtype_code(dtype) == _hip.SSA_F64 else 4)
    return (n + per_line - 1) // per_line * per_line


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def device_info() -> Tuple[int, int, str]:
    lib = load_library()
    cus = ctypes.c_int(0)
    mem = ctypes.c_size_t(0)
    name = ctypes.create_string_buffer(64)
    check(lib.ssa_device_info(ctypes.byref(cus), ctypes.byref(mem), name, 64), "ssa_device_info")
    return cus.value, mem.value, name.value.decode()


# ---------------------------------------------------------------------------------------
def q_assemble(xy: torch.Tensor, w: torch.Tensor, C: torch.Tensor, dtype, *,
               want_Q: bool = True, ld: Optional[int] = None,
               out: Optional[torch.Tensor] = None) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """``MeshOperators.Q_matrix`` (device/mesh.py:435-458) incl. ``q_matrix``
    (distance.py:87-115).  Returns ``(Q [n, ld] or None, qdiag [n] float64)``."""
    lib = load_library()
    n = xy.shape[0]
    qdiag = torch.empty(n, dtype=torch.float64, device=xy.device)
    Q = None
    ldq = 0
    if want_Q:
        ldq = ld or padded_ld(n, dtype)
        Q = out if out is not None else torch.empty((n, ldq), dtype=_tdtype(dtype), device=xy.device)
    check(lib.ssa_q_assemble(ptr(xy), ptr(w), ptr(C), n, ptr(Q), ldq, dtype_code(dtype),
                             ptr(qdiag), current_stream()), "ssa_q_assemble")
    return Q, qdiag


def system_assemble(xy, w, qdiag, Lambda, lap_indptr, lap_indices, lap_data, rows, cols, *,
                    sign: float, dtype, ld: Optional[int] = None, row_scale=None,
                    lower_only: bool = False, alloc_rows: Optional[int] = None) -> torch.Tensor:
    """``_build_system_2d`` / ``_build_system_1d`` (solver/solve_film.py:285-305):
    ``sign * row_scale[rows] * (Q[rows, cols] * w[cols] - Lambda[cols] * Del2[rows, cols])`` as
    ``[nr, ld]``; ``lower_only`` writes only the entries on / below the diagonal."""
    lib = load_library()
    n = xy.shape[0]
    nr = n if rows is None else rows.shape[0]
    nc = cols.shape[0]
    ldo = ld or (padded_ld(nc, dtype) if nc > 1 else 1)
    out = torch.empty((max(nr, alloc_rows or 0), ldo), dtype=_tdtype(dtype), device=xy.device)
    nbytes = lib.ssa_system_assemble_workspace_bytes(n, nr, nc)
    ws = _ws(nbytes, xy.device)
    check(lib.ssa_system_assemble(ptr(xy), ptr(w), ptr(qdiag), ptr(Lambda), n, ptr(lap_indptr),
                                  ptr(lap_indices), ptr(lap_data), ptr(rows), nr, ptr(cols), nc,
                                  float(sign), ptr(row_scale), int(bool(lower_only)), ptr(out), ldo,
                                  dtype_code(dtype), ptr(ws), nbytes, current_stream()),
          "ssa_system_assemble")
    return out


class CholFactors:
    """Device-resident result of :func:`chol_factor`: ``S = L L^T`` (lower triangle of ``L``).
    ``info`` (LAPACK ?potrf convention) is fetched from the device on first access, so several
    factorizations can be enqueued before the host waits for any of them."""

    def __init__(self, L: torch.Tensor, n: int, aux: torch.Tensor, info_device: torch.Tensor,
                 solve_block: int = 4096):
        self.L, self.n, self.aux, self.info_device = L, n, aux, info_device
        self.solve_block = int(solve_block)   # rows of the solves' diagonal blocks (ssa_chol_*_blk)
        self.dtype = L.dtype
        self._info: Optional[int] = None

    @property
    def info(self) -> int:
        if self._info is None:
            self._info = int(self.info_device.item())
        return self._info

    @property
    def lda(self) -> int:
        return self.L.shape[1]


def fetch_chol_infos(factors: Sequence[CholFactors]) -> None:
    """One device-to-host round trip for the ``info`` of several factorizations (instead of one per access)."""
    todo = [f for f in factors if f._info is None]
    if todo:
        for f, v in zip(todo, torch.cat([f.info_device for f in todo]).cpu().tolist()):
            f._info = int(v)


def chol_padded_n(n: int) -> int:
    """Rows / leading dimension the buffer handed to :func:`chol_factor` must provide."""
    return int(load_library().ssa_chol_padded_n(n))


def chol_factor_batch(systems: Sequence[Tuple[torch.Tensor, int]], solve_block: int = 4096) -> List[CholFactors]:
    """In-place Cholesky of several symmetric positive definite matrices ``(S, n)`` (the films
    of a device) in one interleaved schedule (``ssa_chol_factor_batch_blk``).  ``solve_block``: rows of the
    diagonal blocks of the triangular solves (4096, or 2048 for a factorization that serves few solves: a quarter of
    the block-inverse flops, twice the launches per solve); the factors remember it."""
    import ctypes

    lib = load_library()
    count = len(systems)
    if count == 0:
        return []
    dt = dtype_code(systems[0][0].dtype)
    out = []
    for S, n in systems:
        if S.shape[0] < chol_padded_n(n) or S.shape[1] < chol_padded_n(n):
            raise ValueError("chol_factor needs a buffer padded to chol_padded_n(n) rows and columns.")
        if dtype_code(S.dtype) != dt:
            raise ValueError("chol_factor_batch: all matrices must have the same dtype.")
        info = torch.zeros(1, dtype=torch.int32, device=S.device)
        aux = torch.empty(lib.ssa_chol_aux_bytes(n, dt) // S.element_size(), dtype=S.dtype, device=S.device)
        out.append(CholFactors(S, n, aux, info, solve_block))
    PtrArr, I64Arr = ctypes.c_void_p * count, ctypes.c_int64 * count
    check(lib.ssa_chol_factor_batch_blk(
        count, PtrArr(*[f.L.data_ptr() for f in out]), I64Arr(*[f.n for f in out]),
        I64Arr(*[f.lda for f in out]), PtrArr(*[f.info_device.data_ptr() for f in out]),
        PtrArr(*[f.aux.data_ptr() for f in out]), dt, int(solve_block), current_stream()), "ssa_chol_factor_batch_blk")
    return out


def chol_chain_stream_costs() -> Tuple[List[float], List[int]]:
    """``(microseconds, pipe_group)`` of the panel-chain streams of the current device in the order the
    factorization schedule uses them (``ssa_chol_chain_stream_costs``): what a dependent launch costs on each beside
    the caller's stream, and the group of streams it shares a command-processor pipe with (0 = the caller's).
    Empty before the first factorization."""
    import ctypes

    us, grp = (ctypes.c_double * 64)(), (ctypes.c_int32 * 64)()
    k = min(int(load_library().ssa_chol_chain_stream_costs(us, grp, 64)), 64)
    return [float(us[i]) for i in range(k)], [int(grp[i]) for i in range(k)]


def chol_factor(S: torch.Tensor, n: int, solve_block: int = 4096) -> CholFactors:
    """In-place Cholesky of the symmetric positive definite ``S`` (lower triangle given in the
    leading ``n x n`` part of a ``[chol_padded_n(n), lda >= chol_padded_n(n)]`` buffer)."""
    return chol_factor_batch([(S, n)], solve_block)[0]


def chol_solve(f: CholFactors, B: torch.Tensor) -> torch.Tensor:
    """``L L^T X = B`` in place; ``B [n]`` or ``[n, nrhs]``."""
    lib = load_library()
    dt = dtype_code(f.dtype)
    nrhs = 1 if B.dim() == 1 else B.shape[1]
    nbytes = lib.ssa_chol_solve_workspace_bytes(f.n, nrhs, dt)
    ws = _ws(nbytes, B.device)
    check(lib.ssa_chol_solve_blk(ptr(f.L), f.n, f.lda, ptr(f.aux), ptr(B), nrhs, nrhs, dt, ptr(ws), nbytes,
                                 f.solve_block, current_stream()), "ssa_chol_solve_blk")
    return B


def chol_solve_batch(factors: Sequence[CholFactors], rhs: Sequence[torch.Tensor], padded: bool = False) -> List[torch.Tensor]:
    """``L_i L_i^T x_i = b_i`` in place for several factors with ONE right-hand side each (``b_i [n_i]``), the block
    steps of all solves side by side in one launch each (``ssa_chol_solve_batch``); same dtype for all.
    ``padded``: every ``b_i`` has ``chol_padded_n(n_i)`` elements, zero from ``n_i`` on, and is solved where it is
    (no staging copies); the returned vectors are views of the first ``n_i`` elements."""
    import ctypes

    lib = load_library()
    count = len(factors)
    if count == 0:
        return []
    dt = dtype_code(factors[0].dtype)
    if any(dtype_code(f.dtype) != dt for f in factors) or any(b.dim() != 1 or not b.is_contiguous() for b in rhs):
        raise ValueError("chol_solve_batch: one contiguous vector per factor, all of the same dtype.")
    if any(f.solve_block != factors[0].solve_block for f in factors):
        raise ValueError("chol_solve_batch: all factors must have the same solve_block.")
    for f, b in zip(factors, rhs):
        if b.numel() != (chol_padded_n(f.n) if padded else f.n):
            raise ValueError("chol_solve_batch: a right-hand side has the wrong length.")
    nbytes = [lib.ssa_chol_solve_workspace_bytes(f.n, 1, dt) for f in factors]
    ws = [_ws(nb, rhs[0].device) for nb in nbytes]
    PtrArr, I64Arr, SizeArr = ctypes.c_void_p * count, ctypes.c_int64 * count, ctypes.c_size_t * count
    check(lib.ssa_chol_solve_batch_blk(
        count, PtrArr(*[f.L.data_ptr() for f in factors]), I64Arr(*[f.n for f in factors]),
        I64Arr(*[f.lda for f in factors]), PtrArr(*[f.aux.data_ptr() for f in factors]),
        PtrArr(*[b.data_ptr() for b in rhs]), int(bool(padded)), dt, PtrArr(*[w.data_ptr() for w in ws]),
        SizeArr(*nbytes), factors[0].solve_block, current_stream()), "ssa_chol_solve_batch_blk")
    return [b[:f.n] for f, b in zip(factors, rhs)] if padded else list(rhs)


def gemm_ex(opA: int, opB: int, lower_only: bool, A, B, C, M: int, N: int, K: int,
            alpha: float = 1.0, beta: float = 0.0) -> torch.Tensor:
    lib = load_library()
    check(lib.ssa_gemm_ex(opA, opB, int(lower_only), M, N, K, float(alpha), A.data_ptr(), A.stride(0),
                          B.data_ptr(), B.stride(0), float(beta), C.data_ptr(), C.stride(0),
                          dtype_code(C.dtype), current_stream()), "ssa_gemm_ex")
    return C


# ---------------------------------------------------------------------------------------
@dataclass
class LUFactors:
    """Device-resident result of :func:`lu_factor` (the ``lu_piv`` of the reference)."""

    lu: torch.Tensor        # [n, lda] L\\U, row-major
    n: int
    ipiv: torch.Tensor      # [n] int32, LAPACK row interchanges (0-based)
    perm: torch.Tensor      # [n] int64 gather permutation: LU == A[perm]
    aux: torch.Tensor       # inverses of the diagonal blocks (solve phase)
    info: int               # LAPACK info
    dtype: torch.dtype

    @property
    def lda(self) -> int:
        return self.lu.shape[1]


def lu_factor(A: torch.Tensor, n: int) -> LUFactors:
    """``scipy.linalg.lu_factor`` (solver/solve_film.py:279): in-place on ``A [n, lda]``."""
    return lu_factor_batch([(A, n)])[0]


_lu_streams: dict = {}
_num_cus: dict = {}
LU_PANEL_ROWS_PER_GROUP = 256   # kPanelThreads of csrc/lu.hip: rows per workgroup of the cooperative panel kernel


def _device_cus(dev) -> int:
    if dev.index not in _num_cus:
        cus = ctypes.c_int(0)
        with torch.cuda.device(dev):
            check(load_library().ssa_device_info(ctypes.byref(cus), None, None, 0), "ssa_device_info")
        _num_cus[dev.index] = int(cus.value)
    return _num_cus[dev.index]


def lu_concurrency_groups(orders: Sequence[int], num_cus: int) -> List[List[int]]:
    """Which matrices of a batch may be factored side by side by the pivoting route.  Its exact sub-panel kernel
    (``lu_panel_kernel``) is cooperative: ``ceil(n / 256)`` workgroups of one CU each (132 KB of LDS) that
    spin-wait on one another.  Kernels of several matrices that are resident only in part would wait for
    workgroups that cannot be placed, so matrices share the chip only while the workgroups of ALL their panel
    kernels fit on it together; the rest waits for the next group.  Consecutive runs, order preserved."""
    groups, cur, used = [], [], 0
    for i, n in enumerate(orders):
        need = -(-int(n) // LU_PANEL_ROWS_PER_GROUP)
        if cur and used + need > num_cus:
            groups.append(cur)
            cur, used = [], 0
        cur.append(i)
        used += need
    if cur:
        groups.append(cur)
    return groups


def lu_factor_batch(systems: Sequence[Tuple[torch.Tensor, int]]) -> List[LUFactors]:
    """LU factorizations of several matrices (the films of a device), in place, each on a stream of its
    own: a factorization is a strictly sequential chain of latency-bound panel kernels and MFMA trailing
    updates, so the panels of one matrix run beside the updates of the others (the Cholesky route does the
    same inside one schedule, ``ssa_chol_factor_batch``).  Every matrix has its own workspace; the factors
    do not depend on what they are batched with.  Matrices whose cooperative panel kernels would not all be
    resident together are factored group after group (:func:`lu_concurrency_groups`)."""
    lib = load_library()
    if not systems:
        return []
    dev = systems[0][0].device
    main = torch.cuda.current_stream(dev)
    groups = lu_concurrency_groups([n for _, n in systems], _device_cus(dev))
    width = max(len(g) for g in groups)
    side = _lu_streams.setdefault(dev.index, [])
    while len(side) < width - 1:
        side.append(torch.cuda.Stream(device=dev))
    pending = []
    for group in groups:
        streams = [main] + side[:len(group) - 1]
        for i, st in zip(group, streams):
            A, n = systems[i]
            dt = dtype_code(A.dtype)
            ipiv = torch.empty(n, dtype=torch.int32, device=dev)
            info = torch.zeros(1, dtype=torch.int32, device=dev)
            aux = torch.empty(lib.ssa_lu_aux_bytes(n, dt) // A.element_size(), dtype=A.dtype, device=dev)
            nbytes = lib.ssa_lu_factor_workspace_bytes(n, dt)
            ws = _ws(nbytes, dev)
            pending.append((A, n, ipiv, info, aux, ws, nbytes, dt, st))
        # fork: after the assembly (and the zero-fill of info) on the caller's stream -- and after the join of
        # the previous group, which is what keeps two groups' panel kernels apart
        start = torch.cuda.Event()
        start.record(main)
        members = pending[-len(group):]
        for A, n, ipiv, info, aux, ws, nbytes, dt, st in members:
            if st is not main:
                st.wait_event(start)
                for t in (A, ipiv, info, aux, ws):
                    t.record_stream(st)
            check(lib.ssa_lu_factor(ptr(A), n, A.shape[1], ptr(ipiv), ptr(info), ptr(aux), dt, ptr(ws), nbytes,
                                    st.cuda_stream), "ssa_lu_factor")
        for *_, st in members:
            if st is not main:
                done = torch.cuda.Event()
                done.record(st)
                main.wait_event(done)
    out = []
    for A, n, ipiv, info, aux, ws, nbytes, dt, st in pending:
        ipiv_h = np.ascontiguousarray(ipiv.cpu().numpy())  # synchronises the stream
        info_h = int(info.item())
        if info_h < 0:
            raise _hip.HipLibraryError(
                "ssa_lu_factor: the cooperative panel kernel timed out waiting for a peer workgroup."
            )
        perm_h = np.empty(n, dtype=np.int64)
        check(lib.ssa_lu_pivots_to_permutation(ipiv_h.ctypes.data, n, perm_h.ctypes.data),
              "ssa_lu_pivots_to_permutation")
        perm = torch.from_numpy(perm_h).to(dev)
        out.append(LUFactors(lu=A, n=n, ipiv=ipiv, perm=perm, aux=aux, info=info_h, dtype=A.dtype))
    return out


def lu_padded_n(n: int) -> int:
    """Rows / minimum leading dimension of the buffer :func:`lu_factor_nopivot_batch` factors in place."""
    return int(load_library().ssa_lu_padded_n(n))


def lu_factor_nopivot_batch(systems: Sequence[Tuple[torch.Tensor, int]]) -> List[Optional[LUFactors]]:
    """``scipy.linalg.lu_factor`` of several matrices whose partial pivoting never swaps rows (the London
    systems: strictly diagonally dominant), in one look-ahead schedule (``ssa_lu_factor_nopivot_batch``).
    ``systems``: ``(A [lu_padded_n(n), lda >= lu_padded_n(n)], n)``, factored in place.  An entry of the result
    is ``None`` where LAPACK would have interchanged rows (the buffer is then garbage: assemble the matrix
    again and use :func:`lu_factor`); otherwise the factors and ``ipiv == arange`` are LAPACK's to rounding."""
    import ctypes

    lib = load_library()
    count = len(systems)
    if count == 0:
        return []
    dev = systems[0][0].device
    dt = dtype_code(systems[0][0].dtype)
    items = []
    for A, n in systems:
        npad = lu_padded_n(n)
        if A.shape[0] < npad or A.shape[1] < npad:
            raise ValueError("lu_factor_nopivot_batch needs buffers padded to lu_padded_n(n) rows and columns.")
        if dtype_code(A.dtype) != dt:
            raise ValueError("lu_factor_nopivot_batch: all matrices must have the same dtype.")
        ipiv = torch.empty(npad, dtype=torch.int32, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        aux = torch.empty(lib.ssa_lu_aux_bytes(n, dt) // A.element_size(), dtype=A.dtype, device=dev)
        nbytes = lib.ssa_lu_factor_nopivot_workspace_bytes(n, dt)
        items.append((A, n, ipiv, info, aux, _ws(nbytes, dev), nbytes))
    PtrArr, I64Arr, SzArr = ctypes.c_void_p * count, ctypes.c_int64 * count, ctypes.c_size_t * count
    check(lib.ssa_lu_factor_nopivot_batch(
        count, PtrArr(*[it[0].data_ptr() for it in items]), I64Arr(*[it[1] for it in items]),
        I64Arr(*[it[0].shape[1] for it in items]), PtrArr(*[it[2].data_ptr() for it in items]),
        PtrArr(*[it[3].data_ptr() for it in items]), PtrArr(*[it[4].data_ptr() for it in items]), dt,
        PtrArr(*[it[5].data_ptr() for it in items]), SzArr(*[it[6] for it in items]), current_stream()),
        "ssa_lu_factor_nopivot_batch")
    out: List[Optional[LUFactors]] = []
    for A, n, ipiv, info, aux, ws, nbytes in items:
        info_h = int(info.item())  # synchronises the stream
        if info_h < 0:
            out.append(None)
            continue
        out.append(LUFactors(lu=A, n=n, ipiv=ipiv[:n], perm=torch.arange(n, dtype=torch.int64, device=dev), aux=aux,
                             info=info_h, dtype=A.dtype))
    return out


def lu_solve_permuted(f: LUFactors, B: torch.Tensor) -> torch.Tensor:
    """``scipy.linalg.lu_solve`` (solver/solve_film.py:530) on an already row-permuted rhs
    ``B [n] or [n, nrhs]`` (``B = h[perm]``); solved in place and returned."""
    lib = load_library()
    dt = dtype_code(f.dtype)
    nrhs = 1 if B.dim() == 1 else B.shape[1]
    nbytes = lib.ssa_lu_solve_workspace_bytes(f.n, nrhs, dt)
    ws = _ws(nbytes, B.device)
    check(lib.ssa_lu_solve(ptr(f.lu), f.n, f.lda, ptr(f.aux), ptr(B), nrhs, nrhs, dt, ptr(ws), nbytes,
                           current_stream()), "ssa_lu_solve")
    return B


def lu_solve(f: LUFactors, b: torch.Tensor) -> torch.Tensor:
    """Convenience: permutes ``b`` with torch indexing, then :func:`lu_solve_permuted`."""
    return lu_solve_permuted(f, b[f.perm].contiguous())


# ---------------------------------------------------------------------------------------
def gemv(M: torch.Tensor, nr: int, nc: int, x: torch.Tensor, *, xscale=None, xidx=None,
         y: Optional[torch.Tensor] = None, alpha: float = 1.0, beta: float = 0.0) -> torch.Tensor:
    """BLAS gemv at solver/solve_film.py:503 and :565."""
    lib = load_library()
    if y is None:
        y = torch.empty(nr, dtype=M.dtype, device=M.device)
        beta = 0.0
    ldm = M.shape[1] if M.dim() == 2 else nc
    check(lib.ssa_gemv(ptr(M), nr, nc, ldm, ptr(x), ptr(xscale), ptr(xidx), ptr(y), float(alpha),
                       float(beta), dtype_code(M.dtype), current_stream()), "ssa_gemv")
    return y


def row_scale(x: torch.Tensor, s: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``y = s[:, None] * x`` (``out``: where to write it, same shape as ``x``)."""
    lib = load_library()
    y = torch.empty_like(x) if out is None else out
    nvec = 1 if x.dim() == 1 else x.shape[1]
    check(lib.ssa_row_scale(ptr(x), ptr(s), ptr(y), x.shape[0], nvec, dtype_code(x.dtype),
                            current_stream()), "ssa_row_scale")
    return y


def self_field(xy, w, qdiag, g: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """Matrix-free ``Q @ (w * g)`` (solver/solve_film.py:565)."""
    lib = load_library()
    n = xy.shape[0]
    out = torch.empty_like(g)
    nbytes = lib.ssa_self_field_workspace_bytes(n)
    ws = _ws(nbytes, g.device)
    check(lib.ssa_self_field(ptr(xy), ptr(w), ptr(qdiag), ptr(g), n, ptr(out), float(alpha),
                             dtype_code(g.dtype), ptr(ws), nbytes, current_stream()), "ssa_self_field")
    return out


def self_field_rows(xy, w, qdiag, g: torch.Tensor, rows: torch.Tensor, out: torch.Tensor,
                    alpha: float = 1.0) -> torch.Tensor:
    """``out[rows] = (Q @ (w * g))[rows]`` by the all-pairs sum, other entries of ``out`` untouched."""
    lib = load_library()
    n, nr = xy.shape[0], rows.numel()
    if nr == 0:
        return out
    nbytes = lib.ssa_self_field_workspace_bytes(nr)
    ws = _ws(nbytes, g.device)
    check(lib.ssa_self_field_rows(ptr(xy), ptr(w), ptr(qdiag), ptr(g), n, ptr(rows), nr, ptr(out), float(alpha),
                                  dtype_code(g.dtype), ptr(ws), nbytes, current_stream()), "ssa_self_field_rows")
    return out


def london_field_rows(lap_indptr, lap_indices, lap_data, Lambda, g: torch.Tensor, applied: torch.Tensor,
                      other: Optional[torch.Tensor], rows: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """``out[rows] = (Laplacian (Lambda g) - applied - other)[rows]``: the self field of the film interior
    from the London equation (see ``include/superscreen_hip.h``); ``g [n]`` or ``[n, nvec]``."""
    lib = load_library()
    nvec = 1 if g.dim() == 1 else g.shape[1]
    check(lib.ssa_london_field_rows(ptr(lap_indptr), ptr(lap_indices), ptr(lap_data), ptr(Lambda), ptr(g),
                                    ptr(applied), ptr(other) if other is not None else None, ptr(rows),
                                    rows.numel(), nvec, ptr(out), dtype_code(g.dtype), current_stream()),
          "ssa_london_field_rows")
    return out


def self_field_multi(xy, w, qdiag, g: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """Matrix-free ``Q @ (w * g)`` for ``g [n, nvec]`` (one evaluation of r^-3 per pair for all vectors)."""
    lib = load_library()
    n, nvec = g.shape
    out = torch.empty_like(g)
    nbytes = lib.ssa_pairwise_multi_workspace_bytes(n)
    ws = _ws(nbytes, g.device)
    check(lib.ssa_self_field_multi(ptr(xy), ptr(w), ptr(qdiag), ptr(g), n, nvec, ptr(out), float(alpha),
                                   dtype_code(g.dtype), ptr(ws), nbytes, current_stream()), "ssa_self_field_multi")
    return out


def self_field_multi_rows(xy, w, qdiag, g: torch.Tensor, rows: torch.Tensor, out: torch.Tensor,
                          alpha: float = 1.0) -> torch.Tensor:
    """``out[rows, :] = (Q @ (w * g))[rows, :]`` for ``g [n, nvec]``; other rows of ``out`` untouched."""
    lib = load_library()
    n, nvec, nr = g.shape[0], g.shape[1], rows.numel()
    if nr == 0:
        return out
    nbytes = lib.ssa_pairwise_multi_workspace_bytes(nr)
    ws = _ws(nbytes, g.device)
    check(lib.ssa_self_field_multi_rows(ptr(xy), ptr(w), ptr(qdiag), ptr(g), n, nvec, ptr(rows), nr, ptr(out),
                                        float(alpha), dtype_code(g.dtype), ptr(ws), nbytes, current_stream()),
          "ssa_self_field_multi_rows")
    return out


def biot_savart_multi(src_xy, src_areas, src_J, tgt_xy, dz: float, out: torch.Tensor, *,
                      accumulate: bool, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``biot_savart_film_to_film`` for ``src_J [ns, nvec, 2]`` into ``out [nt, nvec]``; with ``rows`` only
    those target rows are evaluated and written."""
    lib = load_library()
    ns, nt, nvec = src_xy.shape[0], tgt_xy.shape[0], out.shape[1]
    if rows is not None:
        nr = rows.numel()
        nbytes = lib.ssa_pairwise_multi_workspace_bytes(max(nr, 1))
        ws = _ws(nbytes, out.device)
        check(lib.ssa_biot_savart_multi_rows(ptr(src_xy), ptr(src_areas), ptr(src_J), ns, ptr(tgt_xy), nt, ptr(rows),
                                             nr, float(dz), nvec, ptr(out), int(bool(accumulate)),
                                             dtype_code(out.dtype), ptr(ws), nbytes, current_stream()),
              "ssa_biot_savart_multi_rows")
        return out
    nbytes = lib.ssa_pairwise_multi_workspace_bytes(nt)
    ws = _ws(nbytes, out.device)
    check(lib.ssa_biot_savart_multi(ptr(src_xy), ptr(src_areas), ptr(src_J), ns, ptr(tgt_xy), nt, float(dz), nvec,
                                    ptr(out), int(bool(accumulate)), dtype_code(out.dtype), ptr(ws), nbytes,
                                    current_stream()), "ssa_biot_savart_multi")
    return out


def film_rhs(applied, other, ha_eff, idx, nvec: int = 1) -> torch.Tensor:
    """``h = Hz[indices] - Ha_eff[indices]`` (solver/solve_film.py:486-488, 526-529)."""
    lib = load_library()
    ni = idx.shape[0]
    shape = (ni,) if nvec == 1 and applied.dim() == 1 else (ni, nvec)
    h = torch.empty(shape, dtype=applied.dtype, device=applied.device)
    check(lib.ssa_film_rhs(ptr(applied), ptr(other), ptr(ha_eff), ptr(idx), ni, nvec, ptr(h),
                           dtype_code(applied.dtype), current_stream()), "ssa_film_rhs")
    return h


def scatter_add(g, idx, gf, nvec: int = 1) -> None:
    """``g[indices] += gf`` (solver/solve_film.py:531)."""
    lib = load_library()
    check(lib.ssa_scatter_add(ptr(g), ptr(idx), ptr(gf), idx.shape[0], nvec, dtype_code(g.dtype),
                              current_stream()), "ssa_scatter_add")


def index_add_scalar(g, idx, values) -> None:
    """``g[hole_indices] += I_circ`` (solver/solve_film.py:498-502)."""
    lib = load_library()
    vals = np.atleast_1d(np.asarray(values, dtype=np.float64))
    check(lib.ssa_index_add_scalar(ptr(g), ptr(idx), idx.shape[0], vals.ctypes.data, len(vals),
                                   dtype_code(g.dtype), current_stream()), "ssa_index_add_scalar")


def current_density(indptr, indices, gx_data, gy_data, g, nvec: int = 1) -> torch.Tensor:
    """``J = [grad_y @ g, -(grad_x @ g)].T`` (solver/solve_film.py:556); float64 output."""
    lib = load_library()
    n = indptr.shape[0] - 1
    shape = (n, 2) if nvec == 1 and g.dim() == 1 else (n, nvec, 2)
    J = torch.empty(shape, dtype=torch.float64, device=g.device)
    check(lib.ssa_current_density(ptr(indptr), ptr(indices), ptr(gx_data), ptr(gy_data), ptr(g), n,
                                  nvec, ptr(J), dtype_code(g.dtype), current_stream()),
          "ssa_current_density")
    return J


def scale(x: torch.Tensor, alpha: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = load_library()
    y = torch.empty_like(x) if out is None else out
    check(lib.ssa_scale(ptr(x), ptr(y), float(alpha), x.numel(), dtype_code(x.dtype),
                        current_stream()), "ssa_scale")
    return y


def biot_savart(src_xy, src_areas, src_J, tgt_xy, dz: float, out: torch.Tensor, *,
                accumulate: bool, src_begin: int = 0, src_end: Optional[int] = None) -> torch.Tensor:
    """``biot_savart_film_to_film`` + ``other[film] += ...`` (solver/solve.py:28-73, :508)."""
    lib = load_library()
    ns, nt = src_xy.shape[0], tgt_xy.shape[0]
    src_end = ns if src_end is None else src_end
    nbytes = lib.ssa_biot_savart_workspace_bytes(nt)
    ws = _ws(nbytes, out.device)
    check(lib.ssa_biot_savart(ptr(src_xy), ptr(src_areas), ptr(src_J), ns, src_begin, src_end,
                              ptr(tgt_xy), nt, float(dz), ptr(out), int(bool(accumulate)),
                              dtype_code(out.dtype), ptr(ws), nbytes, current_stream()),
          "ssa_biot_savart")
    return out


def gemm(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int,
         alpha: float = 1.0, beta: float = 0.0) -> torch.Tensor:
    lib = load_library()
    lda = A.stride(0) if A.dim() == 2 else 1
    ldb = B.stride(0) if B.dim() == 2 else 1
    ldc = C.stride(0) if C.dim() == 2 else 1
    check(lib.ssa_gemm(M, N, K, float(alpha), A.data_ptr(), lda, B.data_ptr(), ldb, float(beta),
                       C.data_ptr(), ldc, dtype_code(C.dtype), current_stream()), "ssa_gemm")
    return C


def sheet_field(src_xy: torch.Tensor, src_areas: torch.Tensor, src_J: torch.Tensor, z0: float,
                eval_xyz: torch.Tensor, prefactor: float, vector: bool) -> torch.Tensor:
    """Field of a film's sheet current at arbitrary points (``sources/current.py:13-110``);
    all inputs float64 device tensors; returns ``[np]`` or ``[np, 3]``."""
    lib = load_library()
    np_, ns = eval_xyz.shape[0], src_xy.shape[0]
    out = torch.empty((np_, 3) if vector else (np_,), dtype=torch.float64, device=eval_xyz.device)
    nbytes = lib.ssa_sheet_field_workspace_bytes(np_, int(vector))
    ws = _ws(nbytes, eval_xyz.device)
    check(lib.ssa_sheet_field(ptr(src_xy), ptr(src_areas), ptr(src_J), ns, float(z0), ptr(eval_xyz), np_,
                              float(prefactor), int(vector), ptr(out), ptr(ws), nbytes, current_stream()),
          "ssa_sheet_field")
    return out


def sheet_potential(src_xy: torch.Tensor, src_areas: torch.Tensor, src_J: torch.Tensor, z0: float,
                    eval_xyz: torch.Tensor, prefactor: float) -> torch.Tensor:
    """In-plane vector potential of a film's sheet current at arbitrary points
    (``solution.py:833-934``): ``[np, 2]`` float64."""
    lib = load_library()
    np_, ns = eval_xyz.shape[0], src_xy.shape[0]
    out = torch.empty((np_, 2), dtype=torch.float64, device=eval_xyz.device)
    nbytes = lib.ssa_sheet_field_workspace_bytes(np_, 1)
    ws = _ws(nbytes, eval_xyz.device)
    check(lib.ssa_sheet_potential(ptr(src_xy), ptr(src_areas), ptr(src_J), ns, float(z0), ptr(eval_xyz), np_,
                                  float(prefactor), ptr(out), ptr(ws), nbytes, current_stream()),
          "ssa_sheet_potential")
    return out


def mfma_probe(iters: int = 2000) -> float:
    """Sustained FP64 MFMA rate of this GPU in TFLOP/s (register-only instruction stream)."""
    lib = load_library()
    sink = torch.zeros(1, dtype=torch.float64, device="cuda")
    flops = ctypes.c_double(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    check(lib.ssa_mfma_probe(iters, ptr(sink), ctypes.byref(flops), current_stream()), "ssa_mfma_probe")  # warm-up
    e0.record()
    check(lib.ssa_mfma_probe(iters, ptr(sink), ctypes.byref(flops), current_stream()), "ssa_mfma_probe")
    e1.record()
    torch.cuda.synchronize()
    return flops.value / (e0.elapsed_time(e1) * 1e-3) / 1e12


def fill_probe(buf: torch.Tensor) -> None:
    lib = load_library()
    check(lib.ssa_fill_probe(ptr(buf), buf.numel() * buf.element_size(), current_stream()),
          "ssa_fill_probe")
