"""Warm (pre-factorized) self-consistent solves of config H: ms per 10-iteration solve (development aid)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic
device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
model = sc.factorize_model(device=device, current_units="uA")
ts = []
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc.solve(model=model, applied_field=sc.ConstantField(0.3 + i), iterations=10, progress_bar=False)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print(f"{os.environ.get('SSA_LIB_PATH', 'default lib')}: warm 10-iteration solve median {np.median(ts[2:]):.2f} ms (all: {' '.join('%.1f' % t for t in ts)})")
