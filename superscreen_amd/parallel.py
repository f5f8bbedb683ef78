"""Multi-GPU execution of the solver path: one process per GPU, ``torch.distributed`` (backend
``"nccl"`` = RCCL over xGMI on MI355X nodes; ``"gloo"`` in the CPU tests).

The reference is single-process; this decomposition is new:

* **Applied-field sweeps, circulating-current scans, mutual-inductance columns** are independent
  solves on one factorized model (``solve`` never mutates it, ``solver/solve.py:391-399``):
  :func:`shard_range` gives each rank a contiguous slice of the sweep; there is NO collective in
  the data path (weak scaling).  Every rank factorizes its own replica (about 0.15 s per 20k-film
  on MI355X, cheaper than shipping 6.6 GB of LU over xGMI).

* **Inter-film Biot-Savart coupling** (``solver/solve.py:499-515``) is a sum over sources, so it is
  split by SOURCE SLICE: for every ordered (source, target) film pair, rank r evaluates the
  sources ``shard_range(n_src, r, world)`` with ``ssa_biot_savart(src_begin, src_end)`` into a
  zero-initialised concatenated field vector; ONE ``all_reduce(SUM)`` per Jacobi iteration
  (sum_f n_f values, 0.97 MB for the 4 x 30k stack) completes it.  xGMI is point-to-point, so the
  single fused buffer (one latency-bound ring pass) is preferred over one collective per film.
  Per-film factor/solve work is replicated on every rank (a film's dense LU does not shard
  naturally), so the 1 -> N gain is bounded by the coupling share.

* **Owner-computes film placement** (:class:`FilmPlacement`, SURVEY.md section 8e "films of one
  device"): film f lives on rank ``f mod world``; a rank assembles, factors and solves only its own
  films and evaluates the complete coupling field of its own TARGET films.  After every pass the
  owners broadcast their films' small result vectors (g, J, self-field, coupling field:
  5 n values, 1 MB for a 25k-vertex film), so every rank holds every iterate and returns the same
  Solutions; nothing of size n^2 ever moves.  With at least as many films as ranks, factorization
  and solve time divide by the number of ranks.
"""
from __future__ import annotations

import itertools
from typing import Callable, Dict, List, Optional, Sequence, Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced ``[begin, end)`` slice of ``range(n)`` for ``rank``."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"Invalid rank {rank} for world size {world}.")
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_list(items: Sequence, rank: int, world: int) -> List:
    b, e = shard_range(len(items), rank, world)
    return list(items[b:e])


def _dist():
    import torch.distributed as dist

    return dist


class CouplingPlan:
    """Distributes the inter-film coupling sums of one Jacobi iteration over the ranks of a
    process group and completes them with a single all-reduce.

    ``pair_kernel(src, tgt, begin, end, out)`` must ADD into ``out`` the field at film ``tgt`` due
    to sources ``[begin, end)`` of film ``src``.  The default (``None``) is the HIP kernel
    ``ssa_biot_savart`` on the model's device-resident data; the CPU tests inject the oracle.
    """

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None,
                 pair_kernel: Optional[Callable] = None):
        dist = _dist()
        if rank is None or world is None:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed is not initialised.")
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world, self.group = rank, world, group
        self.pair_kernel = pair_kernel

    # -- pure bookkeeping (unit-tested without any process group) --------------------------
    @staticmethod
    def tasks(films: Sequence[str], sizes: Dict[str, int], rank: int, world: int):
        """``(src, tgt, begin, end)`` work items of ``rank``: every ordered pair, this rank's
        source slice (ordering = ``itertools.product(films, repeat=2)``, ``solve.py:499``)."""
        out = []
        for src, tgt in itertools.product(films, repeat=2):
            if src == tgt:
                continue
            b, e = shard_range(sizes[src], rank, world)
            if e > b:
                out.append((src, tgt, b, e))
        return out

    # -- execution --------------------------------------------------------------------------
    def reduce_fields(self, films: Sequence[str], other: Dict[str, "object"]) -> None:
        """In-place SUM all-reduce of all films' partial fields as ONE flat buffer."""
        import torch

        if self.world == 1:
            return
        flat = torch.cat([other[f].reshape(-1) for f in films])
        _dist().all_reduce(flat, op=_dist().ReduceOp.SUM, group=self.group)
        off = 0
        for f in films:
            k = other[f].numel()
            other[f].copy_(flat[off:off + k].view_as(other[f]))
            off += k

    def accumulate(self, model, results, other_d) -> None:
        """Called by :func:`superscreen_amd.solver.solve` once per iteration: fills
        ``other_d[film]`` (zero-initialised) with the complete field from all other films."""
        films = list(model.device.films)
        sizes = {f: model.film_data[f].n for f in films}
        kernel = self.pair_kernel or self._hip_pair_kernel(model, results)
        for src, tgt, b, e in self.tasks(films, sizes, self.rank, self.world):
            kernel(src, tgt, b, e, other_d[tgt])
        self.reduce_fields(films, other_d)

    @staticmethod
    def _hip_pair_kernel(model, results):
        from . import kernels

        def run(src, tgt, b, e, out):
            s, t = model.film_data[src], model.film_data[tgt]
            dz = model.film_info[tgt].z0 - model.film_info[src].z0
            kernels.biot_savart(s.xy, s.w_t, results[src].J, t.xy, dz, out, accumulate=True,
                                src_begin=b, src_end=e)

        return run


class FilmPlacement:
    """Owner-computes placement of the films of a coupled stack (see the module docstring).

    Pass it to :func:`superscreen_amd.factorize_model` (the rank then factors only its films) and to
    :func:`superscreen_amd.solve`.  The process group must already exist; tensors travel with
    ``torch.distributed.broadcast`` (RCCL on GPUs)."""

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None):
        dist = _dist()
        if rank is None or world is None:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed is not initialised.")
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"Invalid rank {rank} for world size {world}.")
        self.rank, self.world, self.group = rank, world, group

    def owners(self, films: Sequence[str]) -> Dict[str, int]:
        """``{film: owning rank}``: round-robin in device order."""
        return {f: i % self.world for i, f in enumerate(films)}

    def mine(self, films: Sequence[str]) -> List[str]:
        own = self.owners(films)
        return [f for f in films if own[f] == self.rank]

    def _global_rank(self, group_rank: int) -> int:
        dist = _dist()
        if self.group is None or not hasattr(dist, "get_global_rank"):
            return group_rank
        return dist.get_global_rank(self.group, group_rank)

    def share(self, films: Sequence[str], tensors: Dict[str, Dict[str, "object"]],
              shapes: Dict[str, Dict[str, tuple]], dtypes: Dict[str, Dict[str, "object"]], device) -> None:
        """Completes ``tensors[film][key]`` on every rank: the owner broadcasts, the others receive
        into freshly allocated tensors of the given shape / dtype.  Deterministic order."""
        import torch

        own = self.owners(films)
        for f in films:
            bucket = tensors.setdefault(f, {})
            for key in sorted(shapes[f]):
                if own[f] != self.rank:
                    bucket[key] = torch.empty(shapes[f][key], dtype=dtypes[f][key], device=device)
                if self.world > 1:
                    _dist().broadcast(bucket[key], src=self._global_rank(own[f]), group=self.group)


def solve_sweep_sharded(model, applied_fields: Sequence, *, rank: Optional[int] = None, world: Optional[int] = None,
                        group=None, summarize: Optional[Callable] = None, solve_fn: Optional[Callable] = None,
                        **sweep_kwargs):
    """Applied-field scan over the ranks of a process group (BASELINE config 4): rank r solves the
    contiguous slice ``shard_range(len(applied_fields), r, world)`` with
    :func:`superscreen_amd.solve_sweep` on its own GPU and its own replica of ``model`` -- there is no
    collective in the data path.  Returns ``(begin, end, local)`` with ``local[k]`` = the result of field
    ``begin + k``.

    ``summarize(solutions_of_one_field) -> small picklable value`` (e.g. a fluxoid, a susceptibility):
    if given, the summaries of ALL fields, in field order, are exchanged with one ``all_gather_object``
    and returned as a fourth item, so that every rank holds the complete curve.
    ``solve_fn`` replaces ``solve_sweep`` (the CPU tests inject the oracle)."""
    dist = _dist()
    if rank is None or world is None:
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised.")
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    if solve_fn is None:
        from .sweep import solve_sweep as solve_fn
    begin, end = shard_range(len(applied_fields), rank, world)
    local = solve_fn(model, list(applied_fields[begin:end]), **sweep_kwargs) if end > begin else []
    if summarize is None:
        return begin, end, local
    mine = [summarize(item) for item in local]
    if world == 1:
        return begin, end, local, mine
    gathered = [None] * world
    dist.all_gather_object(gathered, mine, group=group)
    return begin, end, local, [value for part in gathered for value in part]
