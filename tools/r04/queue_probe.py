"""Which part of a 4-film pass depends on GPU_MAX_HW_QUEUES?  Kernel loops timed with events after a 4-film
factorization (development aid)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import superscreen_amd as sc
from superscreen_amd import kernels, synthetic
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fact = (sys.argv[2] != "nofact") if len(sys.argv) > 2 else True
device = synthetic.make_stack_device(100, ("disk",) * nf, z_spacing=0.5, solve_dtype="float64")
if fact:
    model = sc.factorize_model(device=device, current_units="uA")
    fds = [model.film_data[f] for f in device.films]
    xy, w = fds[0].xy, fds[0].w_t
else:
    m = device.meshes[list(device.films)[0]]
    xy = torch.from_numpy(m.sites).cuda()
    w = torch.from_numpy(m.operators.weights).cuda()
n = xy.shape[0]
J = torch.randn(n, 2, dtype=torch.float64, device="cuda")
out = torch.zeros(n, dtype=torch.float64, device="cuda")
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) * 1e3 / reps
a = timed(lambda: kernels.biot_savart(xy, w, J, xy, 0.5, out, accumulate=True), 60)
res = [f"biot_savart {a[0]:.3f} ms (wall {a[1]:.3f})"]
if fact:
    facs = [model.film_systems[f].chol for f in device.films]
    rhs = [torch.randn(kernels.chol_padded_n(f.n), dtype=torch.float64, device="cuda") for f in facs]
    b = timed(lambda: kernels.chol_solve_batch(facs, [r.clone() for r in rhs], padded=True), 20)
    res.append(f"chol_solve_batch x{nf} {b[0]:.3f} ms (wall {b[1]:.3f})")
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} films={nf} fact={fact}: " + " | ".join(res))
