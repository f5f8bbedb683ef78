"""Factorization time of a single-film disk (config 2 at K = 129), median of cold repeats (development aid)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic
K = int(sys.argv[1]) if len(sys.argv) > 1 else 129
device = synthetic.make_stack_device(K, ("disk",), solve_dtype="float64")
ts = []
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA")
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
    del model
print(f"single disk K={K}: factorize median {np.median(ts[2:]):.1f} ms (all: {' '.join('%.1f' % t for t in ts)})")
