"""Where does the float32 error of the GPU path come from?  First pass of the two-film device (no coupling) in
float32 through the Cholesky and the LU route against the float64 result, by mesh size (development aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import superscreen_amd as sc
from superscreen_amd import synthetic

def rel(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b)) / np.max(np.abs(b)))

for K in (26, 45, 64, 91):
    ref = None
    row = []
    for dtype, method in (("float64", "auto"), ("float32", "auto"), ("float32", "lu")):
        device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype=dtype)
        model = sc.factorize_model(device=device, current_units="uA", method=method)
        sols = sc.solve(model=model, applied_field=sc.ConstantField(0.3), iterations=2, progress_bar=False)
        g = {nm: [s.film_solutions[nm].stream for s in sols] for nm in device.films}
        if ref is None:
            ref = g
            continue
        row.append(f"{method}: " + " ".join(f"{nm} {max(rel(a, b) for a, b in zip(g[nm], ref[nm])):.2e}" for nm in g))
        del model, sols
    print(f"K={K}: float32 vs float64 | " + " | ".join(row), flush=True)
