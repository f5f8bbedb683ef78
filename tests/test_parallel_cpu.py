"""N > 1 path on CPU: world_size-2 ``gloo`` process groups exercise the sharding and the single
fused all-reduce of the coupling field.  The per-tile arithmetic is the CPU oracle here (test
infrastructure); on GPUs the same plan drives ``ssa_biot_savart`` slices (tests/test_solve_gpu.py
checks that source slices + accumulate reproduce the full sum bit for bit)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

from superscreen_amd.parallel import CouplingPlan, shard_list, shard_range  # noqa: E402


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 25117):
        for world in (1, 2, 3, 8):
            pieces = [shard_range(n, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
            sizes = [e - b for b, e in pieces]
            assert max(sizes) - min(sizes) <= 1
    assert shard_list(list(range(64)), 3, 8) == list(range(24, 32))
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_coupling_tasks_cover_every_pair_once():
    films = ["a", "b", "c"]
    sizes = {"a": 10, "b": 7, "c": 1}
    for world in (1, 2, 4):
        cover = {}
        for rank in range(world):
            for src, tgt, b, e in CouplingPlan.tasks(films, sizes, rank, world):
                cover.setdefault((src, tgt), []).append((b, e))
        assert set(cover) == {(s, t) for s in films for t in films if s != t}
        for (src, _), slices in cover.items():
            slices.sort()
            assert slices[0][0] == 0 and slices[-1][1] == sizes[src]
            assert all(x[1] == y[0] for x, y in zip(slices, slices[1:]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    import superscreen_oracle as orc
    import torch.distributed as dist

    from superscreen_amd import synthetic
    from superscreen_amd.parallel import CouplingPlan, shard_list

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # three coaxial films with random sheet currents (same seed on every rank)
        rng = np.random.default_rng(7)
        meshes, J, areas, z0 = {}, {}, {}, {}
        for i, K in enumerate((6, 7, 5)):
            name = f"film{i}"
            sites, elements, _ = synthetic.ring_disk_mesh(K)
            meshes[name] = sites
            J[name] = rng.standard_normal((len(sites), 2))
            areas[name] = rng.uniform(0.5, 1.5, len(sites))
            z0[name] = 0.4 * i
        films = list(meshes)

        def pair_kernel(src, tgt, b, e, out):
            H = orc.biot_savart_film_to_film(
                film1_sites=meshes[src][b:e], film1_z0=z0[src], film1_areas=areas[src][b:e],
                film1_J=J[src][b:e], film2_sites=meshes[tgt], film2_z0=z0[tgt])
            out += torch.from_numpy(H)

        plan = CouplingPlan(pair_kernel=pair_kernel)
        assert (plan.rank, plan.world) == (rank, world)
        other = {f: torch.zeros(len(meshes[f]), dtype=torch.float64) for f in films}
        sizes = {f: len(meshes[f]) for f in films}
        for src, tgt, b, e in plan.tasks(films, sizes, plan.rank, plan.world):
            pair_kernel(src, tgt, b, e, other[tgt])
        plan.reduce_fields(films, other)
        # serial reference (solver/solve.py:499-515)
        worst = 0.0
        for tgt in films:
            ref = np.zeros(len(meshes[tgt]))
            for src in films:
                if src != tgt:
                    ref += orc.biot_savart_film_to_film(
                        film1_sites=meshes[src], film1_z0=z0[src], film1_areas=areas[src],
                        film1_J=J[src], film2_sites=meshes[tgt], film2_z0=z0[tgt])
            worst = max(worst, float(np.max(np.abs(other[tgt].numpy() - ref)) / np.max(np.abs(ref))))
        # solve_sweep_sharded: contiguous slices, one all_gather_object for the summaries, field order kept
        from superscreen_amd.parallel import solve_sweep_sharded

        scan = list(np.linspace(0.1, 6.4, 13))               # 13 fields on 2 ranks: 7 + 6
        calls = []

        def fake_sweep(model, fields, iterations=0):
            calls.append(list(fields))
            return [[("solution", model, f, it) for it in range(iterations + 1)] for f in fields]

        b, e, local, curve = solve_sweep_sharded("model", scan, solve_fn=fake_sweep, iterations=2,
                                                 summarize=lambda sols: round(10 * sols[-1][2], 6))
        assert (b, e) == ((0, 7) if rank == 0 else (7, 13)) and len(local) == e - b and len(calls) == 1
        assert calls[0] == scan[b:e] and all(len(s) == 3 for s in local)
        assert curve == [round(10 * f, 6) for f in scan]
        b2, e2, local2 = solve_sweep_sharded("model", scan[:1], solve_fn=fake_sweep)   # fewer fields than ranks
        assert (e2 - b2, len(local2)) == ((1, 1) if rank == 0 else (0, 0))
        # sweep sharding: every field value is solved exactly once across the ranks
        mine = shard_list(list(np.linspace(0.1, 6.4, 64)), rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        q.put((rank, worst, sum(len(g) for g in gathered), sorted(sum(gathered, [])) ==
               sorted(np.linspace(0.1, 6.4, 64).tolist())))
    finally:
        dist.destroy_process_group()


def test_coupling_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, total, complete in out:
        assert worst < 1e-13, (rank, worst)
        assert total == 64 and complete


def test_film_placement_bookkeeping():
    from superscreen_amd.parallel import FilmPlacement

    films = ["a", "b", "c", "d", "e"]
    for world in (1, 2, 3, 8):
        owned = []
        for rank in range(world):
            p = FilmPlacement(rank=rank, world=world)
            assert p.owners(films) == {f: i % world for i, f in enumerate(films)}
            owned += p.mine(films)
        assert sorted(owned) == films          # every film exactly once
    with pytest.raises(ValueError):
        FilmPlacement(rank=2, world=2)


def test_film_placement_helper_groups_bookkeeping():
    """More ranks than films: one group of world // n_films consecutive ranks per film; the first rank of a group
    owns the film, all ranks of the group split the sources of the coupling sums whose target is that film."""
    from superscreen_amd.parallel import FilmPlacement

    films = ["a", "b", "c", "d"]
    for world, gsize in ((8, 2), (9, 2), (12, 3), (16, 4)):
        owned, cover = [], {f: [] for f in films}
        for rank in range(world):
            p = FilmPlacement(rank=rank, world=world, n_films=4, make_groups=False)
            assert p.group_size == gsize
            assert p.owners(films) == {f: i * gsize for i, f in enumerate(films)}
            owned += p.mine(films)
            tg = p.coupling_targets(films)
            if rank >= 4 * gsize:                      # ranks beyond the groups idle
                assert tg == [] and p.mine(films) == [] and p.source_slice(3, 103) == (3, 103)
                continue
            assert tg == [films[rank // gsize]]
            cover[tg[0]].append(p.source_slice(3, 103))
        assert owned == films
        for pieces in cover.values():                  # the slices of a group tile [3, 103) exactly once
            pieces.sort()
            assert pieces[0][0] == 3 and pieces[-1][1] == 103 and all(x[1] == y[0] for x, y in zip(pieces, pieces[1:]))
    # fewer ranks than 2 x films, or n_films not given: the round-robin placement, no groups
    for world in (4, 7):
        p = FilmPlacement(rank=1, world=world, n_films=4, make_groups=False)
        assert p.group_size == 1 and p.coupling_targets(films) == p.mine(films) and p.source_slice(0, 10) == (0, 10)
    with pytest.raises(ValueError):
        FilmPlacement(rank=0, world=8, n_films=4, make_groups=False).owners(films[:3])
    # a helper-group placement without its film's process group must refuse to reduce (the world group would add the
    # partial fields of DIFFERENT films together); without helper groups there is nothing to reduce
    import torch

    with pytest.raises(RuntimeError, match="own group"):
        FilmPlacement(rank=0, world=8, n_films=4, make_groups=False).reduce_coupling([torch.zeros(3)])
    FilmPlacement(rank=1, world=4, n_films=4, make_groups=False).reduce_coupling([torch.zeros(3)])


def _helper_worker(rank, world, port, q):
    """BASELINE config 5 on 8 ranks (gloo, CPU): the Jacobi loop of solve.py:491-536 with the CPU oracle standing in
    for the kernels -- owners solve their film, every rank of a film's group evaluates a source slice of the film's
    coupling field, one all-reduce inside the group, one flat all-reduce across the groups per pass."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    import superscreen_oracle as orc
    import torch.distributed as dist
    from matplotlib.path import Path

    from superscreen_amd import synthetic
    from superscreen_amd.parallel import FilmPlacement

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        K, iterations = 7, 3
        sites, elements, dr = synthetic.ring_disk_mesh(K)
        mesh = orc.make_mesh(sites, elements)
        in_film = Path(synthetic.circle_points((synthetic.film_rings(K) + 0.5) * dr), closed=True).contains_points(sites)
        films = [orc.make_film(f"disk{i}", mesh, z0=0.5 * i, Lambda=0.1, in_film=in_film) for i in range(4)]
        names = [f.name for f in films]
        by_name = {f.name: f for f in films}
        conv = orc.field_conversion_mT_to_uA_per_um()
        applied = {f.name: 1.3 * conv * np.ones(len(sites)) for f in films}
        placement = FilmPlacement(n_films=len(films))
        assert placement.group_size == world // 4 and placement.film_group is not None
        mine, targets = placement.mine(names), placement.coupling_targets(names)
        assert len(targets) == 1 and (mine == targets if rank % placement.group_size == 0 else mine == [])
        n = len(sites)
        shapes = {f: {"g": (n,), "J": (n, 2), "other": (n,)} for f in names}
        dtypes = {f: {"g": torch.float64, "J": torch.float64, "other": torch.float64} for f in names}
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda *a, **kw: (calls.append(kw.get("group")), real(*a, **kw))[1]

        def one_pass(other):
            payload = {f: {} for f in names}
            for f in mine:     # the owner solves its film (solve_film.py:440-574)
                sol = orc.solve_film(by_name[f], applied[f], field_conversion=conv,
                                     field_from_other_films=None if other is None else other[f].numpy())
                payload[f] = {"g": torch.from_numpy(sol.stream.copy()), "J": torch.from_numpy(sol.current_density.copy()),
                              "other": other[f] if other is not None else torch.zeros(n, dtype=torch.float64)}
            placement.share(names, payload, shapes, dtypes, torch.device("cpu"))
            return payload

        trace = [one_pass(None)]
        for _ in range(iterations):
            prev = trace[-1]
            other = {f: torch.zeros(n, dtype=torch.float64) for f in names}
            for src in names:
                for tgt in targets:
                    if src == tgt:
                        continue
                    b, e = placement.source_slice(0, n)
                    other[tgt] += torch.from_numpy(orc.biot_savart_film_to_film(
                        film1_sites=sites[b:e], film1_z0=by_name[src].z0, film1_areas=by_name[src].weights[b:e],
                        film1_J=prev[src]["J"].numpy()[b:e], film2_sites=sites, film2_z0=by_name[tgt].z0))
            placement.reduce_coupling([other[t] for t in targets])
            trace.append(one_pass(other))
        dist.all_reduce = real
        # per pass: one flat all-reduce across the groups; per iteration one more inside this rank's group
        ok = calls.count(None) == iterations + 1 and calls.count(placement.film_group) == iterations
        ok = ok and len(calls) == 2 * iterations + 1
        ref = orc.solve(films, 1.3, iterations=iterations)
        worst = 0.0
        for got, want in zip(trace, ref):
            for f in names:
                for a, b in ((got[f]["g"].numpy(), want[f].stream), (got[f]["J"].numpy(), want[f].current_density)):
                    worst = max(worst, float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
        for f in names:   # the coupling field of the last iterate, complete on every rank
            a, b = trace[-1][f]["other"].numpy() / conv, ref[-1][f].field_from_other_films   # (solve_film.py:566-574)
            worst = max(worst, float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
        q.put((rank, bool(ok), worst))
    finally:
        dist.destroy_process_group()


def test_film_placement_helper_groups_world8_gloo():
    """Config 5 with more ranks than films (4 films, 8 ranks): equal to the single-process Jacobi loop to 1e-12, with
    one cross-group and one in-group collective per pass."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_helper_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, worst in out:
        assert ok, (rank, out)
        assert worst < 1e-12, (rank, worst)


def _share_worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist

    from superscreen_amd.parallel import FilmPlacement

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        films = ["a", "b", "c"]                       # 3 films on 2 ranks: rank 0 owns a and c
        n = {"a": 11, "b": 7, "c": 5}
        gen = torch.Generator().manual_seed(3)        # same numbers on every rank
        full = {f: {"g": torch.randn(n[f], generator=gen, dtype=torch.float32),
                    "J": torch.randn(n[f], 2, generator=gen, dtype=torch.float64),
                    "other": torch.randn(n[f], generator=gen, dtype=torch.float32)} for f in films}
        shapes = {f: {k: tuple(v.shape) for k, v in full[f].items()} for f in films}
        dtypes = {f: {k: v.dtype for k, v in full[f].items()} for f in films}
        placement = FilmPlacement()
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
        payload = {f: ({k: v.clone() for k, v in full[f].items()} if f in placement.mine(films) else {})
                   for f in films}
        placement.share(films, payload, shapes, dtypes, torch.device("cpu"))
        dist.all_reduce = real
        ok = len(calls) == 1                         # ONE collective per pass
        for f in films:
            for k, v in full[f].items():
                got = payload[f][k]
                ok = ok and got.dtype == v.dtype and got.shape == v.shape and torch.equal(got, v)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_film_placement_share_is_one_collective_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_share_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in out), out


def test_sweep_grid_layout():
    from superscreen_amd.parallel import SweepGrid

    assert SweepGrid.layout(2, 8) == (2, 4) and SweepGrid.layout(2, 1) == (1, 1) and SweepGrid.layout(4, 8) == (4, 2)
    assert SweepGrid.layout(2, 5) == (2, 2)                     # one rank idles
    for n_films, world, n_fields in ((2, 8, 64), (2, 4, 13), (3, 8, 64), (2, 5, 7), (1, 4, 10)):
        cover = {}
        for rank in range(world):
            g = SweepGrid(n_films, rank=rank, world=world, make_groups=False)
            if not g.active:
                assert g.field_range(n_fields) == (0, 0) and g.placement is None
                continue
            assert g.rank == g.shard * g.film_ranks + g.film_slot and g.placement.world == g.film_ranks
            b, e = g.field_range(n_fields)
            cover.setdefault(g.film_slot, []).append((b, e))
        for slot, pieces in cover.items():                      # every film slot sees every field exactly once
            pieces.sort()
            assert pieces[0][0] == 0 and pieces[-1][1] == n_fields
            assert all(x[1] == y[0] for x, y in zip(pieces, pieces[1:]))


def _grid_worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist

    from superscreen_amd.parallel import SweepGrid, solve_sweep_grid

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        films = ["washer0", "disk1"]
        grid = SweepGrid(len(films))                 # world 4: two shards of two film owners
        ok = (grid.film_ranks, grid.shards) == (2, 2) and grid.shard == rank // 2 and grid.film_slot == rank % 2
        fields = [0.1 * (k + 1) for k in range(9)]   # 9 fields on 2 shards: 5 + 4
        n = {"washer0": 6, "disk1": 4}
        factored, exchanged = [], []

        def fake_factorize(device, current_units, placement):
            mine = placement.mine(films)
            factored.append(mine)
            return ("model", tuple(mine))

        def fake_sweep(model, my_fields, placement, iterations=0):
            # one "pass": every film's [n, nvec] array is produced by its owner only and completed by ONE all-reduce
            # inside the shard's group (the collective solve_sweep issues through FilmPlacement.share)
            nvec = len(my_fields)
            mine = placement.mine(films)
            tensors = {f: {"g": torch.tensor([[100.0 * films.index(f) + i + v for v in my_fields] for i in range(n[f])],
                                             dtype=torch.float64)} for f in mine}
            shapes = {f: {"g": (n[f], nvec)} for f in films}
            dtypes = {f: {"g": torch.float64} for f in films}
            calls = []
            real = dist.all_reduce
            dist.all_reduce = lambda *a, **kw: (calls.append(kw.get("group")), real(*a, **kw))[1]
            placement.share(films, tensors, shapes, dtypes, torch.device("cpu"))
            dist.all_reduce = real
            exchanged.append(calls)
            return [{f: tensors[f]["g"][:, k].clone() for f in films} for k in range(nvec)]

        b, e, local, model = solve_sweep_grid("device", fields, grid, factorize_fn=fake_factorize, solve_fn=fake_sweep,
                                              iterations=3)
        ok = ok and (b, e) == ((0, 5) if grid.shard == 0 else (5, 9)) and len(local) == e - b
        ok = ok and factored == [[films[grid.film_slot]]] and model == ("model", (films[grid.film_slot],))
        ok = ok and len(exchanged) == 1 and len(exchanged[0]) == 1 and exchanged[0][0] is grid.subgroup
        for k, per_film in enumerate(local):         # every rank holds every film of its shard's fields
            for f in films:
                want = torch.tensor([100.0 * films.index(f) + i + fields[b + k] for i in range(n[f])], dtype=torch.float64)
                ok = ok and torch.equal(per_film[f], want)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sweep_grid_world4_gloo():
    """BASELINE config 4 on a (film owner) x (field shard) grid, world size 4 on gloo: a rank factors ONE film, the
    two shards split the fields, and the result arrays travel in one all-reduce per pass INSIDE a shard's group."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in out), out
