import os, sys, time, cProfile, pstats
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import superscreen_amd as sc
from superscreen_amd import synthetic, kernels
device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
for _ in range(2):
    m = sc.factorize_model(device=device, current_units="uA"); torch.cuda.synchronize(); del m
class Stop(Exception): pass
orig = kernels.system_assemble
def sa(*a, **k): raise Stop()
kernels.system_assemble = sa
pr = cProfile.Profile()
ts=[]
for i in range(20):
    t0=time.perf_counter()
    pr.enable()
    try:
        sc.factorize_model(device=device, current_units="uA")
    except Stop:
        pass
    pr.disable()
    ts.append(time.perf_counter()-t0)
    torch.cuda.synchronize()
print("prologue ms (with cProfile overhead)", np.median(ts)*1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
