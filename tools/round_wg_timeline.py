"""Inside ONE fused round launch of a Cholesky factorization: when each workgroup ran and as what (development aid).

    python tools/round_wg_timeline.py [round=3] [extra SSA_CHOL_DEBUG settings]

Runs config H's factorization with SSA_CHOL_DEBUG=wgtime=<round> (csrc/chol.hip), reads the {role, start, end} triples the
workgroups of that round wrote and prints, per role (1 diagonal block, 2 tile, 3 panel rows, 4 strip tile): count, first
start, last end, median duration, and a 20-us histogram of how many workgroups of each role were running."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402

rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
extra = sys.argv[2] if len(sys.argv) > 2 else ""
device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
path = "/tmp/ssa_wgtime.bin"
for rep in range(3):
    if rep == 2:
        os.environ["SSA_CHOL_DEBUG"] = ",".join(x for x in (f"wgtime={rnd}", extra) if x)
        os.environ["SSA_CHOL_TRACE_FILE"] = path
    elif extra:
        os.environ["SSA_CHOL_DEBUG"] = extra
    model = sc.factorize_model(device=device, current_units="uA")
    torch.cuda.synchronize()
    del model
raw = np.fromfile(path, dtype=np.uint64).reshape(-1, 3)
raw = raw[raw[:, 0] > 0]
t0 = raw[:, 1].min()
us = lambda x: (x.astype(np.int64) - int(t0)) / 100.0
names = {1: "diag", 2: "tile", 3: "panel rows", 4: "strip tile"}
end = us(raw[:, 2]).max()
print(f"round {rnd} {extra}: {len(raw)} workgroups, launch span {end:.0f} us")
for role in (1, 2, 3, 4):
    r = raw[raw[:, 0] == role]
    if len(r) == 0:
        continue
    s, e = us(r[:, 1]), us(r[:, 2])
    print(f"  {names[role]:11s} {len(r):5d} workgroups: first start {s.min():7.1f}  last start {s.max():7.1f}  first end {e.min():7.1f}  "
          f"last end {e.max():7.1f}  median duration {np.median(e - s):6.1f} us")
step = 20.0
print("  running workgroups per 20 us   " + "  ".join(f"{names[k]:>10s}" for k in (1, 2, 3, 4)))
for b in np.arange(0.0, end + step, step):
    row = []
    for role in (1, 2, 3, 4):
        r = raw[raw[:, 0] == role]
        s, e = us(r[:, 1]), us(r[:, 2])
        row.append(int(np.sum((s < b + step) & (e > b))))
    print(f"  {b:7.0f} us                     " + "  ".join(f"{v:10d}" for v in row))
