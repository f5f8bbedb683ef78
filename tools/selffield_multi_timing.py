"""Exterior-row self field of the 11 iterates of a config-H solve: 11 single-vector launches against one multi-vector
launch (development aid)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic, kernels

device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
model = sc.factorize_model(device=device, current_units="uA")
for name in device.films:
    fd = model.film_data[name]
    system = model.film_systems[name]
    ext = np.setdiff1d(np.arange(fd.n, dtype=np.int64), system.indices)
    rows = torch.from_numpy(ext).to(fd.device)
    for nvec in (1, 11):
        g = torch.randn(fd.n, nvec, dtype=torch.float64, device=fd.device)
        out1 = torch.zeros(fd.n, nvec, dtype=torch.float64, device=fd.device)
        outm = torch.zeros(fd.n, nvec, dtype=torch.float64, device=fd.device)
        cols = [g[:, v].contiguous() for v in range(nvec)]
        outs = [torch.zeros(fd.n, dtype=torch.float64, device=fd.device) for _ in range(nvec)]
        def single():
            for v in range(nvec):
                kernels.self_field_rows(fd.xy, fd.w, fd.qdiag, cols[v], rows, outs[v])
        def multi():
            kernels.self_field_multi_rows(fd.xy, fd.w, fd.qdiag, g, rows, outm)
        for fn, label in ((single, "single-vector launches"), (multi, "one multi-vector launch")):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            print(f"{name}: n={fd.n} exterior rows={len(ext)} nvec={nvec}: {label} {1e3 * np.median(ts):.3f} ms")
        ref = torch.stack(outs, dim=1)
        err = (ref[rows] - outm[rows]).abs().max().item() / ref[rows].abs().max().item()
        print(f"   max rel diff multi vs single: {err:.2e}")
