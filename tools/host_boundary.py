"""Host time at the two ends of a cold config-H step (development aid): from the call of factorize_model to its
first kernel launch, and from the last pass's enqueue to the return of solve()."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic, kernels, solver

device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
marks = {}
orig_sa = kernels.system_assemble
orig_patch = solver._patch_exterior_self_fields


def sa(*a, **k):
    marks.setdefault("first_assemble", time.perf_counter())
    return orig_sa(*a, **k)


def patch(*a, **k):
    marks["patch_begin"] = time.perf_counter()
    r = orig_patch(*a, **k)
    marks["patch_end"] = time.perf_counter()
    return r


kernels.system_assemble = sa
solver._patch_exterior_self_fields = patch
rows = []
for i in range(7):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA")
    t1 = time.perf_counter()
    sols = sc.solve(model=model, applied_field=sc.ConstantField(0.3 + i), iterations=10, progress_bar=False)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append((marks["first_assemble"] - t0, t1 - t0, marks["patch_begin"] - t0, marks["patch_end"] - marks["patch_begin"],
                 t2 - marks["patch_end"], t2 - t0, t3 - t2))
    del model, sols
r = 1e3 * np.median(np.array(rows[2:]), axis=0)
print(f"factorize_model: first assembly launch after {r[0]:.2f} ms, returns after {r[1]:.2f} ms")
print(f"solve: exterior self-field patch begins at {r[2]:.2f} ms, takes {r[3]:.2f} ms, solve returns {r[4]:.2f} ms later; step {r[5]:.2f} ms (+{r[6]:.2f} ms sync)")
