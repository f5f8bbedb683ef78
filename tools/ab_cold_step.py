"""A/B of the cold step of config H inside one process: factorize_model(expected_passes=None) (4096-row solve blocks)
against expected_passes=iterations + 1 (2048-row blocks), taking turns (development aid).
    python tools/ab_cold_step.py [K=91] [iterations=10] [reps=12]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
iterations = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")
times = {None: [], iterations + 1: []}
parts = {None: [], iterations + 1: []}
for rep in range(reps + 2):
    for ep in times:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model = sc.factorize_model(device=device, current_units="uA", expected_passes=ep)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sols = sc.solve(model=model, applied_field=sc.ConstantField(0.1 * (rep + 1)), field_units="mT",
                        iterations=iterations, progress_bar=False)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        del model, sols
        if rep >= 2:
            times[ep].append((t2 - t0) * 1e3)
            parts[ep].append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
for ep in times:
    f = np.median([p[0] for p in parts[ep]])
    s = np.median([p[1] for p in parts[ep]])
    print(f"expected_passes={ep}: cold step median {np.median(times[ep]):7.2f} ms (factorize {f:6.2f} + solve {s:6.2f}; with a "
          f"host wait between them)   min {min(times[ep]):7.2f}")
