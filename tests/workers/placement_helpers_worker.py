"""Worker of tests/test_solve_gpu.py::test_film_placement_helper_groups_four_ranks: launched four times by
``torch.distributed.run``; all ranks share cuda:0 (gloo carries the collectives).  Two films on four ranks:
two groups of (owner, helper) -- the layout BASELINE config 5 takes on 8 GPUs (4 films, groups of 2)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402
from superscreen_amd.parallel import FilmPlacement  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    device = synthetic.make_stack_device(10, ("washer", "disk"), z_spacing=0.7)
    films = list(device.films)
    kw = dict(applied_field=sc.ConstantField(0.8), field_units="mT", iterations=4)
    circ = {"hole0": 1.5}
    placement = FilmPlacement(n_films=len(films))
    assert placement.group_size == world // 2 == 2 and placement.owners(films) == {"washer0": 0, "disk1": 2}
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents=circ, placement=placement)
    mine = placement.mine(films)
    assert set(model.film_systems) == set(mine) and (len(mine) == 1) == (rank % 2 == 0), (rank, list(model.film_systems))
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(k.get("group")), real(*a, **k))[1]
    sols = sc.solve(model=model, placement=placement, **kw)
    dist.all_reduce = real
    # per pass one flat all-reduce across the groups, per iteration one inside the film's group
    assert calls.count(None) == 5 and calls.count(placement.film_group) == 4 and len(calls) == 9, calls
    ref_model = sc.factorize_model(device=device, current_units="uA", circulating_currents=circ)
    ref = sc.solve(model=ref_model, **kw)
    assert len(sols) == len(ref) == 5
    worst = 0.0
    for a, b in zip(sols, ref):
        for name in films:
            fa, fb = a.film_solutions[name], b.film_solutions[name]
            for x, y in ((fa.stream, fb.stream), (fa.current_density, fb.current_density), (fa.self_field, fb.self_field)):
                worst = max(worst, float(np.max(np.abs(x - y)) / np.max(np.abs(y))))
            if fb.field_from_other_films is not None:
                x, y = fa.field_from_other_films, fb.field_from_other_films
                worst = max(worst, float(np.max(np.abs(x - y)) / np.max(np.abs(y))))
    assert worst < 1e-12, worst
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: helper groups == single process (max rel diff {worst:.1e}), films {mine}")


if __name__ == "__main__":
    main()
