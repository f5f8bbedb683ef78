"""Worker of tests/test_solve_gpu.py::test_nccl_backend_single_rank: launched by ``torch.distributed.run``
with ONE rank and the ``nccl`` backend (= RCCL) -- what a 1-GPU box can run, since RCCL refuses two ranks
on one device.  The coupling plan's fused all-reduce, the placement's exchange, the sharded sweep and the
C-ABI communicator bootstrap all execute on RCCL; results must equal the plain single-process solve."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402
from superscreen_amd.parallel import CouplingPlan, FilmPlacement, RcclCommunicator, solve_sweep_sharded  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    rank, world = dist.get_rank(), dist.get_world_size()
    t = torch.arange(8, dtype=torch.float64, device="cuda")
    dist.all_reduce(t)                                              # an RCCL kernel really runs
    torch.cuda.synchronize()
    assert torch.equal(t, world * torch.arange(8, dtype=torch.float64, device="cuda"))
    device = synthetic.make_stack_device(10, ("washer", "disk", "disk", "washer"), z_spacing=0.5)
    circ = {"hole0": 1.5, "hole3": -0.5}
    kw = dict(applied_field=sc.ConstantField(0.8), field_units="mT", iterations=3)
    ref_model = sc.factorize_model(device=device, current_units="uA", circulating_currents=circ)
    ref = sc.solve(model=ref_model, **kw)
    plan = sc.solve(model=ref_model, coupling=CouplingPlan(), **kw)            # torch.distributed all-reduce
    comm = RcclCommunicator.from_torch_group()                                 # C-ABI communicator
    plan_c = sc.solve(model=ref_model, coupling=CouplingPlan(comm=comm), **kw)
    placement = FilmPlacement()
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents=circ, placement=placement)
    placed = sc.solve(model=model, placement=placement, **kw)
    for other in (plan, plan_c, placed):
        assert len(other) == len(ref)
        for a, b in zip(other, ref):
            for name in device.films:
                assert np.array_equal(a.film_solutions[name].stream, b.film_solutions[name].stream)
    b, e, local, curve = solve_sweep_sharded(ref_model, [0.1, 0.2, 0.3], iterations=2, all_iterations=False,
                                             summarize=lambda sols: float(sols[-1].film_solutions["disk1"].stream.min()))
    assert (b, e) == (0, 3) and len(local) == 3 and len(curve) == 3
    assert abs(curve[2] / curve[0] - 3.0) > 0                                  # finite numbers came back
    comm.destroy()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: nccl (RCCL) world {world}: coupling plan, C-ABI communicator, placement and sharded sweep ok")


if __name__ == "__main__":
    main()
