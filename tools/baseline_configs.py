"""BASELINE.json configs 2 and 3 through the public API on one MI355X.

config 2: single-film disk, K = 129 (n = 50 311, n_i = 41 419): Q assembly, fused system assembly,
          factorization, one solve_film pass.
config 3: washer + shield disk, K = 81 (n = 19 927 per film): fixed 10 iterations, and iterations until
          max|dg| / max|g| < 1e-8.
python tools/baseline_configs.py [2] [3]       (development aid; bench.py is the contract)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import kernels, synthetic  # noqa: E402

which = set(sys.argv[1:]) or {"2", "3"}


def wall(fn, reps=3):
    out, ts = None, []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return out, min(ts)


if "2" in which:
    device = synthetic.make_stack_device(129, ("disk",), solve_dtype="float64")
    n = len(device.meshes["disk0"].sites)
    model = sc.factorize_model(device=device, current_units="uA")   # warm-up: allocations, lazy init
    fd, system = model.film_data["disk0"], model.film_systems["disk0"]
    ni = len(system.indices)
    del model
    ld = kernels.padded_ld(n, "float64")
    Q = torch.empty((n, ld), dtype=torch.float64, device="cuda")
    C = torch.from_numpy(device.meshes["disk0"].operators.C).cuda()
    _, tq = wall(lambda: kernels.q_assemble(fd.xy, fd.w, C, "float64", out=Q, ld=ld), reps=5)
    del Q
    S, ta = wall(lambda: kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, system.indices_device,
                                                 system.indices_device, sign=1.0, dtype="float64", row_scale=fd.w,
                                                 lower_only=True))
    del S, fd, system
    model, tf = wall(lambda: sc.factorize_model(device=device, current_units="uA"), reps=2)
    sols, ts = wall(lambda: sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=0))
    g = sols[-1].film_solutions["disk0"].stream
    print(f"config 2: single-film disk n={n}, n_i={ni}, float64")
    print(f"  Q assembly (dense Q, {n * n * 8 / 1e9:.2f} GB written): {tq * 1e3:.2f} ms = {n * n * 8 / tq / 1e12:.2f} TB/s")
    print(f"  fused system assembly (lower tiles of S, {ni * ni * 4 / 1e9:.2f} GB): {ta * 1e3:.2f} ms")
    print(f"  factorize_model (host set-up + Q diagonal + assembly + Cholesky, {ni ** 3 / 3 / 1e12:.1f} TFLOP): "
          f"{tf * 1e3:.0f} ms = {ni ** 3 / 3 / tf / 1e12:.1f} TFLOP/s overall")
    print(f"  one solve (rhs, triangular solves, J, self field, Solution on the host): {ts * 1e3:.2f} ms")
    print(f"  cold solve: {1 / (tf + ts):.2f} /s; warm solve: {1 / ts:.0f} /s; min g = {g.min():.6f}, finite: {np.isfinite(g).all()}")
    del model

if "3" in which:
    device = synthetic.make_stack_device(81, ("washer", "disk"), solve_dtype="float64")
    n = len(device.meshes["washer0"].sites)
    model = sc.factorize_model(device=device, current_units="uA")
    del model
    model, tf = wall(lambda: sc.factorize_model(device=device, current_units="uA"))
    sols, t10 = wall(lambda: sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10))
    conv, tc = wall(lambda: sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=200, tolerance=1e-8))
    print(f"config 3: washer + shield disk, n={n}/film, unknowns={[len(s.indices) for s in model.film_systems.values()]}")
    print(f"  factorize_model: {tf * 1e3:.0f} ms; 10 fixed iterations: {t10 * 1e3:.1f} ms -> cold {1 / (tf + t10):.2f} solves/s, "
          f"warm {1 / t10:.1f} /s")
    print(f"  to max|dg|/max|g| < 1e-8: {len(conv) - 1} iterations, {tc * 1e3:.1f} ms -> cold {1 / (tf + tc):.2f} solves/s")
