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
