"""Cold self-consistent solve of an N-film stack (BASELINE config 5 on one GPU: 4 films x ~30k vertices).
python tools/stack_case.py [K=99] [films=4] [z_spacing=1.5]      (development aid)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 99
nfilms = int(sys.argv[2]) if len(sys.argv) > 2 else 4
z = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
kinds = tuple("washer" if i % 2 == 0 else "disk" for i in range(nfilms))
device = synthetic.make_stack_device(K, kinds, z_spacing=z, solve_dtype="float64")
n = len(next(iter(device.meshes.values())).sites)
model = sc.factorize_model(device=device, current_units="uA")  # first call: HBM allocation, lazy init
sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=1)
torch.cuda.synchronize()
del model
times = []
for _ in range(3):
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    times.append((t1 - t0, t2 - t1))
f, s = min(times, key=sum)
unknowns = [len(sy.indices) for sy in model.film_systems.values()]
flops = sum(u ** 3 / 3 for u in unknowns)
print(f"{nfilms} films, K={K}: n={n}/film, unknowns={unknowns}")
print(f"factorize {1e3 * f:.0f} ms ({flops / f / 1e12:.1f} TFLOP/s over the whole factorization), "
      f"11 passes + 10 coupling rounds {1e3 * s:.0f} ms -> {1 / (f + s):.2f} cold solves/s; "
      f"HBM peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
g = [sol.film_solutions[kinds[0] + "0"].stream for sol in sols]
step = [float(np.abs(g[i + 1] - g[i]).max() / np.abs(g[i + 1]).max()) for i in range(len(g) - 1)]
print("Jacobi step sizes max|dg|/max|g|:", " ".join(f"{x:.1e}" for x in step))
fields = [0.1 * (k + 1) for k in range(16)]
sc.solve_sweep(model, fields[:4], iterations=10, all_iterations=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
sc.solve_sweep(model, fields, iterations=10, all_iterations=False)
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"16-field sweep on the factorized stack: {1e3 * t:.0f} ms -> {16 / t:.0f} solves/s")
