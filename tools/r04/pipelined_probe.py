"""bench.py's pipelined cold solves with the solve stream at normal / high priority (development aid)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import superscreen_amd as sc
from superscreen_amd import synthetic
device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
orig = torch.cuda.Stream
calls = [0]
def make(*a, **k):
    calls[0] += 1
    if calls[0] == 2 and prio:      # the second stream bench creates is the solve stream
        return orig(*a, priority=-1, **k)
    return orig(*a, **k)
torch.cuda.Stream = make
out = bench.pipelined_cold_solves(sc, torch, device, 10, steps=10)
print(f"solve stream priority {'high' if prio else 'normal'}: {out['pipelined_cold_solves_per_s']:.2f} solves/s ({out['pipelined_cold_solve_ms']:.1f} ms)")
