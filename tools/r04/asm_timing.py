"""System (A) assembly alone: the reference's full A (solve_film.py:296-305) and the lower triangle of diag(w) A the
Cholesky route consumes, float64 and float32, 4 back-to-back calls between two events (development aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import superscreen_amd as sc
from superscreen_amd import kernels, synthetic

for K in [int(a) for a in sys.argv[1:]] or [91, 129]:
    device = synthetic.make_stack_device(K, ("disk",), solve_dtype="float64")
    model = sc.factorize_model(device=device, current_units="uA")
    name = list(device.films)[0]
    fd, system = model.film_data[name], model.film_systems[name]
    ix = system.indices_device
    ni = len(system.indices)
    del model
    torch.cuda.empty_cache()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for dtype, es in (("float64", 8), ("float32", 4)):
        for lower in (False, True):
            fn = lambda: kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix, ix, sign=1.0 if lower else -1.0,
                                                 dtype=dtype, row_scale=fd.w if lower else None, lower_only=lower)
            out = fn()
            chk = float(out[:ni, :ni].double().tril().abs().sum())
            del out
            ts = []
            for _ in range(5):
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4)
            t = float(np.median(ts))
            bytes_ = ni * ni * es * (0.5 if lower else 1.0)
            print(f"K={K} n_i={ni} {dtype} {'lower' if lower else 'full '}: {t:7.3f} ms  {bytes_ / t / 1e6:7.0f} GB/s   checksum {chk!r}", flush=True)
