"""Factorization time of config H (assembly + factorization of both films), median of N cold repeats.
usage: python tools/fact_timing.py [method] [dtype] [K]   (development aid; A/B of builds: SSA_LIB_PATH=...)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic
method = sys.argv[1] if len(sys.argv) > 1 else "auto"
dtype = sys.argv[2] if len(sys.argv) > 2 else "float64"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 91
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype=dtype)
ts = []
for rep in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA", method=method)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
    del model
print(f"{os.environ.get('SSA_LIB_PATH', 'default lib')}: {method} {dtype} K={K} factorize median {np.median(ts[2:]):.1f} ms  (all: {' '.join('%.1f' % t for t in ts)})")
