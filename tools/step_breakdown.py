"""Wall-clock breakdown of one bench step (development aid)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")


def step():
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1


step()
for _ in range(2):
    a, b = step()
    print(f"factorize {a*1e3:.1f} ms   solve(11 passes) {b*1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
