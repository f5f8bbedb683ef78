"""Per-step wall times of the bench workload in ONE process (is the first process on a fresh box slow?)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

t_start = time.perf_counter()
device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
print(f"device built at {time.perf_counter() - t_start:.1f} s")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i:2d} at {t0 - t_start:6.1f} s: factorize {1e3 * (t1 - t0):7.1f} ms  solve {1e3 * (t2 - t1):6.1f} ms")
