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
  owners' small result vectors (g, J, self-field, coupling field: 5 n values, 1 MB for a 25k-vertex
  film) are exchanged with ONE collective (a sum all-reduce of a flat buffer in which every vector has
  its place and only the owner writes), so every rank holds every iterate and returns the same
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


class RcclCommunicator:
    """An RCCL communicator owned through the C ABI (``ssa_rccl_*``, ``include/superscreen_hip.h`` section 7),
    for callers that want the coupling all-reduce without ``torch.distributed`` in the data path.  One
    process per GPU; rank 0 draws the 128-byte id and every rank passes the same id.

    ``RcclCommunicator.from_torch_group()`` bootstraps from an existing ``torch.distributed`` group (any
    backend: the id travels through ``broadcast_object_list``); ``RcclCommunicator(id, rank, world)`` takes an
    id distributed by other means."""

    def __init__(self, unique_id: bytes, rank: int, world: int):
        import ctypes

        from . import _hip

        lib = _hip.load_library()
        if len(unique_id) != 128:
            raise ValueError("An RCCL unique id has 128 bytes.")
        handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _hip.check(lib.ssa_rccl_comm_create(ctypes.byref(handle), world, rank, buf), "ssa_rccl_comm_create")
        self.handle, self.rank, self.world = handle, rank, world

    @staticmethod
    def new_unique_id() -> bytes:
        import ctypes

        from . import _hip

        buf = ctypes.create_string_buffer(128)
        _hip.check(_hip.load_library().ssa_rccl_unique_id(buf), "ssa_rccl_unique_id")
        return buf.raw

    @classmethod
    def from_torch_group(cls, group=None) -> "RcclCommunicator":
        dist = _dist()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        if world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
        return cls(box[0], rank, world)

    def all_reduce_sum_(self, flat) -> None:
        """In-place sum over the ranks of a contiguous float32 / float64 device tensor, enqueued on the
        current stream (``ssa_coupling_allreduce``)."""
        from . import _hip

        _hip.check(_hip.load_library().ssa_coupling_allreduce(
            _hip.ptr(flat), flat.numel(), _hip.dtype_code(flat.dtype), self.handle, _hip.current_stream()),
            "ssa_coupling_allreduce")

    def destroy(self) -> None:
        from . import _hip

        if self.handle is not None:
            _hip.check(_hip.load_library().ssa_rccl_comm_destroy(self.handle), "ssa_rccl_comm_destroy")
            self.handle = None


class CouplingPlan:
    """Distributes the inter-film coupling sums of one Jacobi iteration over the ranks of a
    process group and completes them with a single all-reduce.

    ``pair_kernel(src, tgt, begin, end, out)`` must ADD into ``out`` the field at film ``tgt`` due
    to sources ``[begin, end)`` of film ``src``.  The default (``None``) is the HIP kernel
    ``ssa_biot_savart`` on the model's device-resident data; the CPU tests inject the oracle.
    ``comm``: an :class:`RcclCommunicator`; the all-reduce then goes through the C ABI
    (``ssa_coupling_allreduce``) instead of ``torch.distributed``.
    """

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None,
                 pair_kernel: Optional[Callable] = None, comm: Optional[RcclCommunicator] = None):
        if comm is not None:
            rank, world = comm.rank, comm.world
        dist = _dist()
        if rank is None or world is None:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed is not initialised.")
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world, self.group = rank, world, group
        self.pair_kernel = pair_kernel
        self.comm = comm

    # -- pure bookkeeping (unit-tested without any process group) --------------------------
    @staticmethod
    def tasks(films: Sequence[str], sizes: Dict[str, int], rank: int, world: int,
              ranges: Optional[Dict[str, Tuple[int, int]]] = None):
        """``(src, tgt, begin, end)`` work items of ``rank``: every ordered pair, this rank's
        source slice (ordering = ``itertools.product(films, repeat=2)``, ``solve.py:499``).
        ``ranges[src] = (lo, hi)``: only the sources ``[lo, hi)`` of a film are split (the vertices that
        can carry a sheet current; the others contribute exact zeros); default: all ``sizes[src]``."""
        out = []
        for src, tgt in itertools.product(films, repeat=2):
            if src == tgt:
                continue
            lo, hi = ranges[src] if ranges is not None else (0, sizes[src])
            b, e = shard_range(hi - lo, rank, world)
            if e > b:
                out.append((src, tgt, lo + b, lo + e))
        return out

    # -- execution --------------------------------------------------------------------------
    def reduce_fields(self, films: Sequence[str], other: Dict[str, "object"]) -> None:
        """In-place SUM all-reduce of all films' partial fields as ONE flat buffer."""
        import torch

        if self.world == 1 and self.comm is None:
            return
        flat = torch.cat([other[f].reshape(-1) for f in films])
        if self.comm is not None:
            self.comm.all_reduce_sum_(flat)
        else:
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
        ranges = {f: getattr(model.film_data[f], "src_range", (0, sizes[f])) for f in films}
        for src, tgt, b, e in self.tasks(films, sizes, self.rank, self.world, ranges):
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
    :func:`superscreen_amd.solve`.  The process group (or the :class:`RcclCommunicator` ``comm``) must
    already exist; the result vectors travel in one sum all-reduce per pass (:meth:`share`).

    **More ranks than films** (``n_films`` given and ``world >= 2 * n_films``; BASELINE config 5 on 8 GPUs): the
    ranks form one group of ``world // n_films`` consecutive ranks per film.  The first rank of a group OWNS the
    film (assembles, factors and solves it -- a film's dense factorization does not shard);
    the others are its HELPERS: every rank of the group evaluates a source slice of the coupling sums whose
    TARGET is the group's film (``solver/solve.py:499-515`` is a sum over sources), and one sum all-reduce INSIDE
    the group (:meth:`reduce_coupling`: n values between xGMI neighbours) completes the field on the owner
    before it solves.  The flat all-reduce of :meth:`share` stays the only collective that crosses groups.
    Ranks beyond ``n_films * (world // n_films)`` idle (they take part in :meth:`share` with zeros).
    Every rank of ``group`` must construct the placement (``torch.distributed.new_group`` is collective);
    ``make_groups=False`` skips that (bookkeeping tests)."""

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None,
                 comm: Optional[RcclCommunicator] = None, n_films: Optional[int] = None, make_groups: bool = True):
        if comm is not None:
            rank, world = comm.rank, comm.world
        dist = _dist()
        if rank is None or world is None:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed is not initialised.")
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"Invalid rank {rank} for world size {world}.")
        self.rank, self.world, self.group, self.comm = rank, world, group, comm
        self.n_films = n_films
        self.group_size = world // n_films if (n_films is not None and n_films >= 1 and world >= 2 * n_films) else 1
        self.film_group = None       # torch.distributed group of this rank's film (owner + helpers)
        if self.group_size > 1:
            self.film_index = rank // self.group_size if rank < n_films * self.group_size else -1
            self.slot = rank % self.group_size if self.film_index >= 0 else -1
            if make_groups:
                for f in range(n_films):                      # collective: every rank creates every group
                    members = [self._global_rank(f * self.group_size + k) for k in range(self.group_size)]
                    g = dist.new_group(ranks=members)
                    if f == self.film_index:
                        self.film_group = g
        else:
            self.film_index, self.slot = -1, 0

    def owners(self, films: Sequence[str]) -> Dict[str, int]:
        """``{film: owning rank}``: round-robin in device order; with helper groups the first rank of each group."""
        if self.group_size > 1:
            self._check(films)
            return {f: i * self.group_size for i, f in enumerate(films)}
        return {f: i % self.world for i, f in enumerate(films)}

    def mine(self, films: Sequence[str]) -> List[str]:
        own = self.owners(films)
        return [f for f in films if own[f] == self.rank]

    def _check(self, films: Sequence[str]) -> None:
        if self.n_films is not None and len(films) != self.n_films:
            raise ValueError(f"This placement was made for {self.n_films} films, the device has {len(films)}.")

    def coupling_targets(self, films: Sequence[str]) -> List[str]:
        """The films whose coupling field (field from the other films) this rank works on: the films it owns, or --
        with helper groups -- the film of its group."""
        if self.group_size > 1:
            self._check(films)
            return [films[self.film_index]] if self.film_index >= 0 else []
        return self.mine(films)

    def source_slice(self, lo: int, hi: int) -> Tuple[int, int]:
        """This rank's share of the sources ``[lo, hi)`` of a coupling sum (all of them without helper groups)."""
        if self.group_size > 1 and self.slot >= 0:
            b, e = shard_range(hi - lo, self.slot, self.group_size)
            return lo + b, lo + e
        return lo, hi

    def reduce_coupling(self, fields: Sequence["object"]) -> None:
        """Sums the partial coupling fields of this rank's film over the ranks of its group, in place (one collective
        for all tensors of ``fields``; nothing to do without helper groups)."""
        import torch

        if self.group_size <= 1 or self.film_index < 0 or not fields:
            return
        if self.film_group is None:
            # (make_groups=False, or a placement built on an RcclCommunicator: without the film's own group the
            # all-reduce would run over the WORLD group and add the partial fields of DIFFERENT films together)
            raise RuntimeError("FilmPlacement with helper groups has no process group for this rank's film "
                               "(constructed with make_groups=False or without torch.distributed): the coupling "
                               "sums of a film can only be reduced inside its own group.")
        if len(fields) == 1:
            _dist().all_reduce(fields[0], op=_dist().ReduceOp.SUM, group=self.film_group)
            return
        flat = torch.cat([t.reshape(-1) for t in fields])
        _dist().all_reduce(flat, op=_dist().ReduceOp.SUM, group=self.film_group)
        off = 0
        for t in fields:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()

    def _global_rank(self, group_rank: int) -> int:
        dist = _dist()
        if self.group is None or not hasattr(dist, "get_global_rank"):
            return group_rank
        return dist.get_global_rank(self.group, group_rank)

    def share(self, films: Sequence[str], tensors: Dict[str, Dict[str, "object"]],
              shapes: Dict[str, Dict[str, tuple]], dtypes: Dict[str, Dict[str, "object"]], device) -> None:
        """Completes ``tensors[film][key]`` on every rank with ONE collective per pass: the vectors of all
        films have fixed places in one flat float64 buffer (float32 values widen exactly); a rank writes
        the vectors of the films it owns and leaves zeros elsewhere, so a single sum all-reduce (x + 0 is
        exact) hands every rank every vector; the other ranks' films are unpacked into fresh tensors of
        the given shape / dtype.  About 5 n values per film (1 MB for a 25k-vertex film): the exchange is
        latency-bound, one collective instead of one broadcast per (film, key) is what counts, and it is
        the same all-reduce the coupling plan uses (``ssa_coupling_allreduce`` with ``self.comm``,
        else ``torch.distributed``: RCCL on GPUs, gloo in the CPU tests)."""
        import math

        import torch

        if self.world == 1:
            return
        own = self.owners(films)
        keys = {f: sorted(shapes[f]) for f in films}
        total = sum(math.prod(shapes[f][k]) for f in films for k in keys[f])
        flat = torch.zeros(total, dtype=torch.float64, device=device)
        off = 0
        for f in films:
            for k in keys[f]:
                cnt = math.prod(shapes[f][k])
                if own[f] == self.rank:
                    flat[off:off + cnt].copy_(tensors[f][k].reshape(-1))
                off += cnt
        if self.comm is not None:
            self.comm.all_reduce_sum_(flat)
        else:
            _dist().all_reduce(flat, op=_dist().ReduceOp.SUM, group=self.group)
        off = 0
        for f in films:
            bucket = tensors.setdefault(f, {})
            for k in keys[f]:
                cnt = math.prod(shapes[f][k])
                if own[f] != self.rank:
                    bucket[k] = flat[off:off + cnt].to(dtypes[f][k]).reshape(shapes[f][k]).contiguous()
                off += cnt


class SweepGrid:
    """Two-dimensional placement of a scan (BASELINE config 4) on the ranks of a node: (film owner) x (field shard).

    A scan on replicas (:func:`solve_sweep_sharded`) does not strong-scale: every rank factors EVERY film and then
    streams every factor for its few columns.  Here the ``world`` ranks form ``shards`` groups of ``film_ranks``
    ranks (``film_ranks`` = min(number of films, world)); inside a group the films are placed owner-computes
    (:class:`FilmPlacement` on the group), so a rank factors and sweeps ONE film (of a two-film device), and the
    group's ranks exchange the ``[n, nvec]`` result arrays with one sum all-reduce per pass; the groups split the
    fields.  Per rank: 1 / film_ranks of the factorization flops and of the solve traffic, 1 / shards of the
    columns.  Rank r -> (shard r // film_ranks, film slot r % film_ranks): the ranks of a group are neighbours.

    Every rank of ``group`` must construct the grid (``torch.distributed.new_group`` is collective)."""

    def __init__(self, n_films: int, rank: Optional[int] = None, world: Optional[int] = None, group=None,
                 make_groups: bool = True):
        if rank is None or world is None:
            dist = _dist()
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed is not initialised.")
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world = rank, world
        self.film_ranks, self.shards = self.layout(n_films, world)
        self.active = rank < self.film_ranks * self.shards     # (world not a multiple of film_ranks: the rest idles)
        self.shard = rank // self.film_ranks if self.active else -1
        self.film_slot = rank % self.film_ranks if self.active else -1
        self.subgroup = None
        if make_groups and self.film_ranks > 1:
            dist = _dist()
            for sh in range(self.shards):                       # collective: every rank creates every group
                members = [self._global(group, sh * self.film_ranks + k) for k in range(self.film_ranks)]
                g = dist.new_group(ranks=members)
                if sh == self.shard:
                    self.subgroup = g
        self.placement = (FilmPlacement(rank=self.film_slot, world=self.film_ranks, group=self.subgroup)
                          if self.active else None)

    @staticmethod
    def layout(n_films: int, world: int) -> Tuple[int, int]:
        """``(film_ranks, shards)`` for ``world`` ranks and ``n_films`` films."""
        if n_films < 1 or world < 1:
            raise ValueError("SweepGrid needs at least one film and one rank.")
        film_ranks = min(n_films, world)
        return film_ranks, world // film_ranks

    @staticmethod
    def _global(group, group_rank: int) -> int:
        dist = _dist()
        if group is None or not hasattr(dist, "get_global_rank"):
            return group_rank
        return dist.get_global_rank(group, group_rank)

    def field_range(self, n_fields: int) -> Tuple[int, int]:
        """The fields of this rank's shard (every rank of a shard's group takes the same slice)."""
        if not self.active:
            return 0, 0
        return shard_range(n_fields, self.shard, self.shards)


def solve_sweep_grid(device, applied_fields: Sequence, grid: SweepGrid, *, current_units: str = "uA", model=None,
                     factorize_fn: Optional[Callable] = None, solve_fn: Optional[Callable] = None, **sweep_kwargs):
    """A scan on a :class:`SweepGrid`: this rank factors the films it owns (``factorize_model(placement=...)``;
    or pass a ``model`` that was factorized with ``grid.placement`` before -- the reference pattern is ONE
    ``factorize_model`` and many ``solve(model=...)``, ``device/device.py:619-627``: the factorization then stays
    out of the scan) and solves its shard's fields with :func:`superscreen_amd.solve_sweep` under the group's
    placement.  Returns ``(begin, end, local, model)``; every rank of a shard's group returns the same ``local``.
    ``factorize_fn`` / ``solve_fn`` replace the library calls (the CPU tests inject stand-ins)."""
    if not grid.active:
        return 0, 0, [], model
    if model is None:
        if factorize_fn is None:
            from .solver import factorize_model as factorize_fn
        model = factorize_fn(device=device, current_units=current_units, placement=grid.placement)
    if solve_fn is None:
        from .sweep import solve_sweep as solve_fn
    begin, end = grid.field_range(len(applied_fields))
    local = solve_fn(model, list(applied_fields[begin:end]), placement=grid.placement, **sweep_kwargs) if end > begin else []
    return begin, end, local, model


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
