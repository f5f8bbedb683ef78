"""Sweeps: many applied fields through ONE factorization, all at once.

The reference answers a field scan by calling ``solve(model=...)`` in a loop (its former
``solve_many`` was removed in v0.9, ``docs/about/changelog.rst:120-124``); each call is
memory-bound (one right-hand side streams the whole factor).  Here the ``nvec`` fields of a scan
are carried as the columns of ``[n, nvec]`` operands through the same steps as ``solve_film``
(``solver/solve_film.py:486-565``) and the Jacobi loop of ``solve`` (``solver/solve.py:491-536``):

* right-hand sides, scatter, sheet current: the ``nvec`` forms of the vector kernels;
* ``gf = lu_solve(lu_piv, h)``: the block triangular solves on the MFMA GEMM (``nrhs = nvec``)
  instead of a GEMV chain -- the factor is read once for all fields;
* self field and inter-film coupling: ``ssa_self_field_multi`` / ``ssa_biot_savart_multi``, which
  evaluate r^-3 once per pair for 16 fields.

Every field's iterates equal those of its own ``solve(model=..., applied_field=field)`` call up to
rounding (different summation order only).  Films with terminals or vortices are not handled here
(their extra terms do not depend on the applied field; use ``solve`` in a loop).
"""
from __future__ import annotations

import itertools
from typing import Callable, List, Optional, Sequence, Union

import numpy as np

from .solution import FilmSolution, Solution
from .sources import ConstantField
from .units import field_conversion_factor


def solve_sweep(model, applied_fields: Sequence[Union[float, Callable]], *, field_units: str = "mT",
                iterations: int = 0, return_solutions: bool = True, all_iterations: bool = True,
                circulating_currents: Optional[Sequence[dict]] = None, placement: Optional[object] = None,
                _solver: str = "superscreen_amd.solve_sweep") -> Optional[List[List[Solution]]]:
    """Self-consistent solutions for every applied field of a scan.

    ``applied_fields``: callables ``f(x, y, z)`` (e.g. :func:`superscreen_amd.ConstantField`) or
    plain numbers (uniform fields in ``field_units``).  Returns ``result[k]`` = the list of
    ``iterations + 1`` Solutions of field ``k`` (1 for a single film), like ``solve`` would.
    ``circulating_currents``: optionally one ``{hole: current}`` dict per column that replaces the
    model's circulating currents for that column (a scan over circulating currents, or the columns
    of a mutual-inductance matrix, ``device/device.py:619-627``); the model is not modified.
    ``all_iterations=False`` keeps only the final iterate (``solve(...)[-1]``): ``result[k]`` then
    has one Solution, and the self field -- an output, not an input of the next iteration
    (``solver/solve_film.py:565-572``) -- is evaluated for the final pass only.
    ``placement``: a :class:`superscreen_amd.parallel.FilmPlacement` (owner-computes): this rank carries only the
    films it factored through the passes, evaluates the coupling field of those target films, and the ranks of
    the placement's group exchange the ``[n, nvec]`` result arrays with ONE sum all-reduce per pass; every rank
    returns the same Solutions.  With :class:`superscreen_amd.parallel.SweepGrid` the ranks of a node form
    (film owner) x (field shard) and a scan strong-scales in both directions."""
    import torch

    from . import _hip, kernels

    _hip.require_gpu()
    device = model.device
    films = list(device.films)
    if any(name in device.terminals for name in films):
        raise NotImplementedError("solve_sweep does not handle films with terminals; loop over solve().")
    if any(info.vortices for info in model.film_info.values()):
        raise NotImplementedError("solve_sweep does not handle vortices; loop over solve().")
    if getattr(model, "method", "auto") == "mixed":
        raise NotImplementedError("solve_sweep does not refine a float32 factorization (method='mixed'); loop over solve().")
    fields = [f if callable(f) else ConstantField(float(f)) for f in applied_fields]
    nvec = len(fields)
    if nvec == 0:
        return [] if return_solutions else None
    current_units = model.current_units
    if circulating_currents is None:
        column_currents = [dict(model.circulating_currents)] * nvec
    else:
        from .units import current_to_float

        if len(circulating_currents) != nvec:
            raise ValueError(f"Expected {nvec} circulating-current dicts (one per field), got {len(circulating_currents)}.")
        column_currents = []
        for currents in circulating_currents:
            for hole_name in currents:
                if hole_name not in device.holes:
                    raise KeyError(f"Unknown hole {hole_name!r}.")
            column_currents.append({k: current_to_float(v, current_units) for k, v in currents.items()})
    uniform_currents = all(c == column_currents[0] for c in column_currents)
    conv = field_conversion_factor(field_units, current_units, length_units=device.length_units)
    dtype = device.solve_dtype
    info_of, fd_of = model.film_info, model.film_data
    if placement is None:
        placement = model.__dict__.get("_placement")
    mine = films if placement is None else placement.mine(films)

    # applied fields on the sites: [n, nvec] per film (host evaluation like solve.py:422-436)
    applied_h, applied_d = {}, {}
    for name in films:
        mesh = device.meshes[name]
        x, y = mesh.sites[:, 0], mesh.sites[:, 1]
        z = info_of[name].z0 * np.ones(len(x))
        cols = []
        for f in fields:
            Hz = np.squeeze(np.asarray(f(x, y, z)) * conv)
            Hz = Hz * np.ones(len(x)) if Hz.ndim == 0 else Hz
            if Hz.ndim != 1:
                raise ValueError(f"Expected applied_field to return a 1D vector, got a {Hz.shape[1]}D vector.")
            cols.append(Hz)
        H = np.ascontiguousarray(np.stack(cols, axis=1).astype(dtype, copy=False))
        applied_h[name] = H
        applied_d[name] = torch.from_numpy(H).to(fd_of[name].device)

    def solve_columns(solve_in_place, rhs):
        """float64, <= 64 fields: the skinny kernel streams the factor once, any column count.  Otherwise
        the block triangular solves run on 128 x 128 MFMA tiles: pad the right-hand sides with zero
        columns to a multiple of 128 so that every tile is a full one (the guarded edge path is several
        times slower than the extra flops cost)."""
        if nvec <= 64 and rhs.dtype == torch.float64:
            return solve_in_place(rhs)
        npad = -(-nvec // 128) * 128
        if npad == nvec:
            return solve_in_place(rhs)
        B = torch.zeros((rhs.shape[0], npad), dtype=rhs.dtype, device=rhs.device)
        B[:, :nvec] = rhs
        return solve_in_place(B)[:, :nvec].contiguous()

    _hole_terms = {}

    def hole_terms(name):
        """``g[hole] += I`` and the holes' effective field (``solve_film.py:498-503``) as ``[n, nvec]``
        arrays: independent of the applied field and of the iteration, evaluated once per distinct set
        of circulating currents."""
        if name not in _hole_terms:
            fd = fd_of[name]
            cols = []
            for k in range(1 if uniform_currents else nvec):
                ha = torch.zeros(fd.n, dtype=fd.tdtype, device=fd.device)
                g1 = torch.zeros(fd.n, dtype=fd.tdtype, device=fd.device)
                for hole_name, hs in model.hole_systems[name].items():
                    if len(hs.indices) == 0:   # a hole without a mesh vertex: an empty system (solver.py)
                        continue
                    kernels.index_add_scalar(g1, hs.indices_device, column_currents[k].get(hole_name, 0))
                    kernels.gemv(hs.A_device, fd.n, len(hs.indices), g1, xidx=hs.indices_device, y=ha,
                                 alpha=-1.0, beta=1.0)
                cols.append((g1, ha))
            if uniform_currents:
                _hole_terms[name] = (cols[0][0][:, None].expand(fd.n, nvec).contiguous(),
                                     cols[0][1][:, None].expand(fd.n, nvec).contiguous())
            else:
                _hole_terms[name] = (torch.stack([c[0] for c in cols], dim=1).contiguous(),
                                     torch.stack([c[1] for c in cols], dim=1).contiguous())
        return _hole_terms[name]

    def run_pass(other_d, want_self_field=True):
        results = {}
        for name in mine:
            fd, info, system = fd_of[name], info_of[name], model.film_systems[name]
            g1_all, ha_all = hole_terms(name)
            g = g1_all.clone()
            other = None if other_d is None else other_d[name]
            has_unknowns = len(system.indices) > 0   # a film without unknowns: empty system, g stays at the hole terms
            if not has_unknowns:
                gf = None
            elif system.chol is not None:
                h = kernels.film_rhs(applied_d[name], other, ha_all, system.indices_device, nvec=nvec)
                gf = solve_columns(lambda B: kernels.chol_solve(system.chol, B),
                                   kernels.row_scale(h, system.neg_w_device))
            else:
                h = kernels.film_rhs(applied_d[name], other, ha_all, system.rhs_indices_device, nvec=nvec)
                gf = solve_columns(lambda B: kernels.lu_solve_permuted(system.factors, B), h)
            if gf is not None:
                kernels.scatter_add(g, system.indices_device, gf, nvec=nvec)
            J = kernels.current_density(*fd.grad, g, nvec=nvec)              # [n, nvec, 2]
            sf = None
            if want_self_field:
                if (model.self_field_mode in ("auto", "london") and fd.tdtype == torch.float64
                        and not info.lambda_info.inhomogeneous and has_unknowns):
                    # interior rows from the London equation, the rest by the all-pairs sum (solver.py)
                    if system.exterior_device is None:
                        exterior = np.setdiff1d(np.arange(fd.n, dtype=np.int64), system.indices)
                        system.exterior_device = torch.from_numpy(exterior).to(fd.device)
                    sf = torch.empty_like(g)
                    kernels.london_field_rows(*fd.lap, fd.Lambda, g, applied_d[name], other, system.indices_device, sf)
                    kernels.self_field_multi_rows(fd.xy, fd.w, fd.qdiag, g, system.exterior_device, sf)
                else:
                    sf = kernels.self_field_multi(fd.xy, fd.w, fd.qdiag, g)
            results[name] = (g, J, sf)
        if placement is not None:   # owner-computes: one collective hands every rank every film's arrays
            results = share_results(results, want_self_field)
        return results

    def share_results(results, with_self_field):
        keys = ("g", "J", "sf") if with_self_field else ("g", "J")
        shapes, dtypes, tensors = {}, {}, {}
        for name in films:
            fd = fd_of[name]
            shapes[name] = {"g": (fd.n, nvec), "J": (fd.n, nvec, 2), "sf": (fd.n, nvec)}
            dtypes[name] = {"g": fd.tdtype, "J": torch.float64, "sf": fd.tdtype}
            shapes[name] = {k: shapes[name][k] for k in keys}
            dtypes[name] = {k: dtypes[name][k] for k in keys}
            if name in results:
                g, J, sf = results[name]
                tensors[name] = {"g": g, "J": J}
                if with_self_field:
                    tensors[name]["sf"] = sf
        placement.share(films, tensors, shapes, dtypes, fd_of[films[0]].device)
        return {name: (tensors[name]["g"], tensors[name]["J"], tensors[name].get("sf")) for name in films}

    def share_coupling(other_d):
        """The coupling fields of the iterate that is returned: every rank needs them for its Solutions."""
        shapes = {name: {"o": (fd_of[name].n, nvec)} for name in films}
        dtypes = {name: {"o": fd_of[name].tdtype} for name in films}
        tensors = {name: {"o": other_d[name]} for name in films if name in other_d}
        placement.share(films, tensors, shapes, dtypes, fd_of[films[0]].device)
        return {name: tensors[name]["o"] for name in films}

    def to_host(results, other_d):
        """Field-major host arrays ([nvec, n(, 2)]): every field's slice is then a contiguous view,
        no per-field host copy.  Transposes and the division by the field conversion run on the device;
        the copies land in pinned memory and are only waited for once, after the last pass."""
        def fetch(t):
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            return h

        out = {}
        for name in films:
            g, J, sf = results[name]
            out[name] = (fetch(g.t().contiguous()), fetch(J.permute(1, 0, 2).contiguous()),
                         fetch((sf / conv).t().contiguous()),
                         None if other_d is None else fetch((other_d[name] / conv).t().contiguous()))
        return out

    trace = []
    n_pass = iterations if (len(films) >= 2 and iterations >= 1) else 0
    results = run_pass(None, all_iterations or n_pass == 0)
    if return_solutions and (all_iterations or n_pass == 0):
        trace.append(to_host(results, None))
    for it in range(n_pass):
        last = it == n_pass - 1
        other_d = {name: torch.zeros((fd_of[name].n, nvec), dtype=fd_of[name].tdtype, device=fd_of[name].device)
                   for name in mine}
        for src, tgt in itertools.product(films, repeat=2):  # solve.py:499-515
            if src == tgt or tgt not in other_d:   # (owner-computes: the coupling field of this rank's target films)
                continue
            s, t = fd_of[src], fd_of[tgt]
            b, e = s.src_range  # vertices that can carry current (solver.FilmDeviceData): the rest adds exact zeros
            # an iterate that is not returned only feeds the next solve: its coupling field is needed on
            # the unknowns' rows of the target film (h = Hz[ix] - ..., solve_film.py:529) and nowhere else
            only_unknowns = None if (all_iterations or last) else model.film_systems[tgt].indices_device
            if only_unknowns is not None and len(model.film_systems[tgt].indices) == 0:
                continue   # nothing of this iterate's coupling field is used
            kernels.biot_savart_multi(s.xy[b:e], s.w_t[b:e], results[src][1][b:e], t.xy,
                                      info_of[tgt].z0 - info_of[src].z0, other_d[tgt], accumulate=True,
                                      rows=only_unknowns)
        results = run_pass(other_d, all_iterations or last)
        if return_solutions and (all_iterations or last):
            trace.append(to_host(results, other_d if placement is None else share_coupling(other_d)))
    if not return_solutions:
        torch.cuda.synchronize()
        return None
    out: List[List[Solution]] = []
    device_copy = device.copy(with_mesh=True, copy_mesh=False)  # what Solution.__init__ would make, once
    torch.cuda.synchronize()                                    # the pinned copies of every pass have landed
    trace = [{name: tuple(None if a is None else a.numpy() for a in arrs) for name, arrs in host.items()}
             for host in trace]
    applied_out = {name: np.ascontiguousarray((applied_h[name] / conv).T) for name in films}
    for k, field in enumerate(fields):
        sols = []
        for host in trace:
            fs = {}
            for name in films:
                g, J, sf, other = host[name]
                fs[name] = FilmSolution(
                    stream=g[k], current_density=J[k], applied_field=applied_out[name][k],
                    self_field=sf[k], field_from_other_films=None if other is None else other[k])
            sols.append(Solution(device=device_copy, film_solutions=fs, applied_field_func=field,
                                 field_units=field_units, current_units=current_units,
                                 circulating_currents=dict(column_currents[k]),
                                 terminal_currents=model.terminal_currents, vortices=model.vortices, solver=_solver,
                                 _device_is_copy=True))
        out.append(sols)
    return out
