"""Where does it hang?  Phases with prints (development aid)."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(80, exit=True)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import superscreen_amd as sc
from superscreen_amd import synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "lu"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
def say(*a):
    print(*a, flush=True)
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")
say("device made")
if which in ("chol_then_lu", "chol"):
    m = sc.factorize_model(device=device, current_units="uA", method="auto"); torch.cuda.synchronize(); say("chol factorized")
    s = sc.solve(model=m, applied_field=sc.ConstantField(0.7), iterations=2); torch.cuda.synchronize(); say("chol solved")
    del m, s
if which in ("chol_then_lu", "lu"):
    m = sc.factorize_model(device=device, current_units="uA", method="lu"); torch.cuda.synchronize(); say("lu factorized")
    s = sc.solve(model=m, applied_field=sc.ConstantField(0.7), iterations=2); torch.cuda.synchronize(); say("lu solved")
say("done")
