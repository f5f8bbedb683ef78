"""One cold solve of a larger two-film device (default K = 128: 49 537 vertices per film), with the
residual of the film systems as a size-independent check (development aid)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import kernels, synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")
n = len(device.meshes["washer0"].sites)
model = sc.factorize_model(device=device, current_units="uA")  # first call: HBM allocation, lazy init
torch.cuda.synchronize()
del model
t0 = time.perf_counter()
model = sc.factorize_model(device=device, current_units="uA")
torch.cuda.synchronize()
t1 = time.perf_counter()
sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"K={K} n={n}/film unknowns={[len(s.indices) for s in model.film_systems.values()]} "
      f"factorize {1e3*(t1-t0):.0f} ms  solve {1e3*(t2-t1):.0f} ms  "
      f"HBM in use {torch.cuda.memory_allocated()/2**30:.1f} GiB (peak {torch.cuda.max_memory_allocated()/2**30:.1f})")
# residual of S x = b for a random b on the bigger film, matrix-free (assemble S again, gemv)
name = "disk1"
system, fd = model.film_systems[name], model.film_data[name]
ni = len(system.indices)
b = torch.randn(ni, dtype=torch.float64, device="cuda")
x = kernels.chol_solve(system.chol, b.clone())
S = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, system.indices_device, system.indices_device,
                            sign=1.0, dtype="float64", row_scale=fd.w)
r = kernels.gemv(S, ni, ni, x) - b
print(f"residual |S x - b| / |b| = {float(r.abs().max() / b.abs().max()):.2e}")
g = sols[-1].film_solutions[name].stream
print("finite:", bool(np.isfinite(g).all()), " max|g| =", float(np.abs(g).max()))
