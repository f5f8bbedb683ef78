"""Factorization time of the 4-film stack (config 5 on one GPU), median of cold repeats (development aid)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic
method = sys.argv[1] if len(sys.argv) > 1 else "auto"
device = synthetic.make_stack_device(100, ("disk",) * 4, solve_dtype="float64")
ts = []
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA", method=method)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
    del model
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}: 4-film stack {method} factorize median {np.median(ts[1:]):.1f} ms ({' '.join('%.1f' % t for t in ts)})")
