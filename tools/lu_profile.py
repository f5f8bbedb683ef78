"""One LU-route factorization of config H (for rocprofv3 --kernel-trace --stats; development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic
device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
for rep in range(3):
    t0 = time.perf_counter()
    model = sc.factorize_model(device=device, current_units="uA", method="lu")
    torch.cuda.synchronize()
    print(f"lu factorize {1e3 * (time.perf_counter() - t0):.1f} ms")
    del model
