"""Throughput of a 64-field scan on config H: solve_sweep vs a loop of warm solve() calls."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")
model = sc.factorize_model(device=device, current_units="uA")
fields = [0.05 * (k + 1) for k in range(nf)]
for tag, ret, every in (("Solutions of every iteration", True, True), ("final Solutions", True, False),
                        ("device only, every iteration's self field", False, True),
                        ("device only, final self field", False, False)):
    sc.solve_sweep(model, fields[:16], iterations=10, return_solutions=ret, all_iterations=every)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc.solve_sweep(model, fields, iterations=10, return_solutions=ret, all_iterations=every)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"solve_sweep {nf} fields, 10 iterations ({tag}): {t*1e3:.1f} ms -> {nf/t:.1f} solves/s")
t0 = time.perf_counter()
for f in fields[:8]:
    sc.solve(model=model, applied_field=sc.ConstantField(f), iterations=10)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 8
print(f"solve() loop: {t*1e3:.1f} ms per field -> {1/t:.1f} solves/s")
