"""Host profile of one solve() call with a single pass (what stands between the factorization and the first pass;
development aid)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc
from superscreen_amd import synthetic
device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
model = sc.factorize_model(device=device, current_units="uA")
for _ in range(3):
    sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=1)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
pr = cProfile.Profile()
pr.enable()
m2 = sc.factorize_model(device=device, current_units="uA")
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(14)
