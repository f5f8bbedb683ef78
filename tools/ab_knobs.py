"""A/B of SSA_CHOL_DEBUG settings (csrc/chol.hip: CholDebug) INSIDE ONE PROCESS on one box: the variants take turns,
repetition by repetition, so that clock drift and the box itself cancel (development aid).

    python tools/ab_knobs.py [--case H|H32|c2|c5] [--reps 9] [--passes 11] "" "early=0" "finish=0" ...

Prints the median factorization time (assembly + factorization of all films, host clock around a synchronised
factorize_model) per variant.  Variants with finish=0 / mirror=0 produce unusable factors: timing only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

args = sys.argv[1:]
case, reps, passes = "H", 9, None
while args and args[0].startswith("--"):
    if args[0] == "--case":
        case = args[1]
    elif args[0] == "--reps":
        reps = int(args[1])
    elif args[0] == "--passes":          # factorize_model(expected_passes=...): 2048-row solve blocks up to 24
        passes = int(args[1])
    args = args[2:]
variants = args or [""]
CASES = {"H": (91, ("washer", "disk"), "float64"), "H32": (91, ("washer", "disk"), "float32"),
         "c2": (129, ("disk",), "float64"), "c5": (100, ("disk",) * 4, "float64"), "c3": (81, ("washer", "disk"), "float64")}
K, kinds, dtype = CASES[case]
device = synthetic.make_stack_device(K, kinds, solve_dtype=dtype)
times = {v: [] for v in variants}
for rep in range(reps + 2):
    for v in variants:
        if v:
            os.environ["SSA_CHOL_DEBUG"] = v
        else:
            os.environ.pop("SSA_CHOL_DEBUG", None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model = sc.factorize_model(device=device, current_units="uA", expected_passes=passes)
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        del model
        if rep >= 2:
            times[v].append(dt)
os.environ.pop("SSA_CHOL_DEBUG", None)
base = float(np.median(times[variants[0]]))
for v in variants:
    t = np.array(times[v])
    print(f"case {case} {v or '(default)':40s} median {np.median(t):7.2f} ms  min {t.min():7.2f}  max {t.max():7.2f}   "
          f"{np.median(t) - base:+6.2f} ms vs first", flush=True)
