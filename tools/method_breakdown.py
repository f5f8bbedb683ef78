"""factorize / solve wall-clock of config H for each factorization route (development aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")
for method in ("auto", "lu"):
    for rep in range(3):
        t0 = time.perf_counter()
        model = sc.factorize_model(device=device, current_units="uA", method=method)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"method={method}: factorize {1e3*(t1-t0):.1f} ms  solve {1e3*(t2-t1):.1f} ms")
    del model
