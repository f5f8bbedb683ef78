import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import superscreen_amd as sc
from superscreen_amd import synthetic
for K, kinds in ((26, ("disk",)), (26, ("washer", "disk")), (45, ("washer", "disk"))):
    device = synthetic.make_stack_device(K, kinds, solve_dtype="float64")
    n = len(next(iter(device.meshes.values())).sites)
    sc.solve(device=device, applied_field=sc.ConstantField(1.0), iterations=5)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = sc.factorize_model(device=device, current_units="uA")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        sc.solve(model=m, applied_field=sc.ConstantField(1.0), iterations=5)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    f, s = min(ts, key=sum)
    print(f"K={K} films={len(kinds)} n={n}: factorize {f*1e3:.2f} ms, solve(5 it) {s*1e3:.2f} ms")
