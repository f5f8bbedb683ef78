"""Repeats the cold factorization of config H (and of a 4-film stack) many times and checks the residual of
the film systems every time: a guard against rare stream-ordering races in the look-ahead schedule."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import kernels, synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for K, kinds in ((91, ("washer", "disk")), (60, ("washer", "disk", "washer", "disk"))):
    device = synthetic.make_stack_device(K, kinds, z_spacing=1.5, solve_dtype="float64")
    worst = 0.0
    ref = None
    for rep in range(reps):
        model = sc.factorize_model(device=device, current_units="uA")
        for name, system in model.film_systems.items():
            fd = model.film_data[name]
            ni = len(system.indices)
            torch.manual_seed(rep)
            b = torch.randn(ni, dtype=torch.float64, device="cuda")
            x = kernels.chol_solve(system.chol, b.clone())
            S = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, system.indices_device,
                                        system.indices_device, sign=1.0, dtype="float64", row_scale=fd.w)
            r = float((kernels.gemv(S, ni, ni, x) - b).abs().max() / b.abs().max())
            worst = max(worst, r)
            assert r < 1e-11, (K, rep, name, r)
        g = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=3)[-1].film_solutions[kinds[0] + "0"].stream
        if ref is None:
            ref = g
        assert (g == ref).all(), (K, rep, "not reproducible")
        del model
    print(f"K={K} films={len(kinds)}: {reps} cold factorizations, worst residual {worst:.2e}, solutions bit-identical")

# routes and precisions interleaved in one process (Cholesky and LU schedules share the chain-stream pool; every change
# of route re-uses streams the other route just left)
device64 = synthetic.make_stack_device(64, ("washer", "disk"), solve_dtype="float64")
device32 = synthetic.make_stack_device(64, ("washer", "disk"), solve_dtype="float32")
first = {}
for rep in range(max(4, reps // 5)):
    for dev, method in ((device64, "auto"), (device32, "auto"), (device64, "lu"), (device32, "lu")):
        model = sc.factorize_model(device=dev, current_units="uA", method=method)
        g = sc.solve(model=model, applied_field=sc.ConstantField(0.7), iterations=2)[-1].film_solutions["disk1"].stream
        key = (dev.solve_dtype, method)
        if key not in first:
            first[key] = g
        assert (g == first[key]).all(), (key, rep, "not reproducible")
        del model
print(f"interleaved routes: {max(4, reps // 5)} x (cholesky f64, cholesky f32, lu f64, lu f32), solutions bit-identical")
