import sys, time, torch
sys.path.insert(0, "/root/repo")
import superscreen_amd as sc
from superscreen_amd import synthetic
for dt in ("float64", "float32"):
    device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype=dt)
    for rep in range(3):
        t0 = time.perf_counter()
        model = sc.factorize_model(device=device, current_units="uA")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(dt, f"factorize {1e3*(t1-t0):.1f} ms  solve {1e3*(t2-t1):.1f} ms", [type(s.chol).__name__ for s in model.film_systems.values()])
