"""Cold / warm solve and sweep timing of config H for the library selected by SSA_LIB_PATH (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic, kernels
K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, out
tf, model = timed(lambda: sc.factorize_model(device=device, current_units="uA"))
ts, sols = timed(lambda: sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10))
tw, _ = timed(lambda: sc.solve_sweep(model, [0.1 * k for k in range(1, 65)], iterations=10, all_iterations=False), reps=3)
t8, _ = timed(lambda: sc.solve_sweep(model, [0.1 * k for k in range(1, 9)], iterations=10, all_iterations=False), reps=3)
g = sols[-1].film_solutions["disk1"].stream
print(f"lib={os.environ.get('SSA_LIB_PATH', 'default')}: factorize {tf:.1f} ms, 11-pass solve {ts:.1f} ms, cold {tf + ts:.1f} ms, "
      f"64-field sweep {tw:.1f} ms, 8-field sweep {t8:.1f} ms, checksum {float(np.abs(g).sum()):.12e}")
n = len(device.meshes["washer0"].sites)
fd = model.film_data["washer0"]
C = torch.from_numpy(device.meshes["washer0"].operators.C).cuda()
ld = kernels.padded_ld(n, "float64")
Q = torch.empty((n, ld), dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for nn in (n,):
    kernels.q_assemble(fd.xy, fd.w, C, "float64", out=Q, ld=ld); ts = []
    for _ in range(7):
        e0.record(); kernels.q_assemble(fd.xy, fd.w, C, "float64", out=Q, ld=ld); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"q_assemble n={nn}: {np.median(ts):.3f} ms -> {nn * nn * 8 / np.median(ts) / 1e6:.0f} GB/s")
