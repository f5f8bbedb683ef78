"""cProfile of the host side of one cold step of config H (development aid)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")


def step():
    model = sc.factorize_model(device=device, current_units="uA")
    sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
    torch.cuda.synchronize()
    return model, sols


for _ in range(2):
    m = step()
    del m
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    m = step()
    ts.append(time.perf_counter() - t0)
    del m
print("step ms", [round(1e3 * t, 1) for t in ts])
t0 = time.perf_counter()
model = sc.factorize_model(device=device, current_units="uA")
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"factorize_model returned after {1e3 * (t1 - t0):.1f} ms (host enqueue), GPU done after {1e3 * (t2 - t0):.1f} ms")
t0 = time.perf_counter()
sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"solve returned after {1e3 * (t1 - t0):.1f} ms")
pr = cProfile.Profile()
pr.enable()
m = step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
