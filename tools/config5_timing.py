"""Cold self-consistent solve of the 4-film stack (config 5 on one GPU): phases (development aid).
usage: [PYTHONPATH=<tree>] python tools/config5_timing.py"""
import os, sys, time
import numpy as np
import torch
if not os.environ.get("PYTHONPATH"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic
print("package:", os.path.dirname(sc.__file__))
device = synthetic.make_stack_device(100, ("disk",) * 4, z_spacing=0.5, solve_dtype="float64")
rows = []
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10, progress_bar=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1))
    del model, sols
r = 1e3 * np.median(np.array(rows[1:]), axis=0)
print(f"4-film stack: factorize {r[0]:.1f} ms, 11 passes {r[1]:.1f} ms, total {r.sum():.1f} ms   (all: {' '.join('%.0f+%.0f' % (1e3*a, 1e3*b) for a, b in rows)})")
