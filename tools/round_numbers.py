"""Factorize / solve wall clock of the BASELINE configurations for both factorization routes and both solve
dtypes on one MI355X (the tables of DESIGN.md section 3; development aid).

    python tools/round_numbers.py [quick]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, out


def case(label, K, kinds, dz=0.5, iterations=10, dtypes=("float64", "float32"), methods=("auto", "lu")):
    for dtype in dtypes:
        device = synthetic.make_stack_device(K, kinds, z_spacing=dz, solve_dtype=dtype)
        for method in methods:
            model = None
            tf, model = timed(lambda: sc.factorize_model(device=device, current_units="uA", method=method))
            ts, sols = timed(lambda: sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=iterations))
            unknowns = [len(s.indices) for s in model.film_systems.values()]
            flops = sum(u ** 3 for u in unknowns) * (1 / 3 if method == "auto" else 2 / 3)
            print(f"{label:34s} {dtype:8s} {method:5s} factorize {tf:7.1f} ms ({flops / tf / 1e9:5.1f} TFLOP/s)  "
                  f"solve({len(sols)} passes) {ts:6.1f} ms  cold {1e3 / (tf + ts):6.2f}/s  unknowns {unknowns}", flush=True)
            del model, sols
            torch.cuda.empty_cache()


quick = len(sys.argv) > 1
case("config H: washer+disk 2x25117", 91, ("washer", "disk"))
case("config 3: washer+disk 2x19927", 81, ("washer", "disk"))
if not quick:
    case("config 2: disk 50311", 129, ("disk",), iterations=0)
    case("config 5: 4 disks 4x30301", 100, ("disk",) * 4, dtypes=("float64",))
    case("config 1: disk 2107", 26, ("disk",), iterations=0, dtypes=("float64",))
