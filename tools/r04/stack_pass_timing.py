"""Warm 10-iteration solves of the 4-film stack (config 5 on one GPU) with a per-phase view of where the host waits
(development aid): total, and with return_solutions=False (no host unpacking)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import superscreen_amd as sc
from superscreen_amd import synthetic
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
device = synthetic.make_stack_device(K, ("disk",) * nf, z_spacing=0.5, solve_dtype="float64")
model = sc.factorize_model(device=device, current_units="uA")
out = []
for ret in (True, False):
    ts = []
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.solve(model=model, applied_field=sc.ConstantField(0.3 + i), iterations=10, progress_bar=False, return_solutions=ret,
                 **({} if ret else {"save_path": None}))
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    out.append(f"return_solutions={ret}: {np.median(ts[2:]):.1f} ms")
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} K={K} films={nf}: " + " | ".join(out))
