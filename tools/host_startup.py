"""Host time of factorize_model up to the factorization call, by function (development aid): the GPU can only
start a film's first panel once the host has launched its assembly."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic, kernels
device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
for _ in range(2):
    m = sc.factorize_model(device=device, current_units="uA"); torch.cuda.synchronize(); del m
orig = kernels.chol_factor_batch
stamp = {}
def wrapped(systems):
    stamp["t"] = time.perf_counter()
    return orig(systems)
kernels.chol_factor_batch = wrapped
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = sc.factorize_model(device=device, current_units="uA")
    torch.cuda.synchronize()
    print(f"host reaches the factorization call after {1e3 * (stamp['t'] - t0):.2f} ms")
    del m
pr = cProfile.Profile()
pr.enable()
m = sc.factorize_model(device=device, current_units="uA")
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
