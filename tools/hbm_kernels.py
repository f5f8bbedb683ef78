"""The two HBM-bound kernels of the path on config H's films, for the PMC passes of tools/collect_profiles.py:
five dense-Q assemblies (ssa_q_assemble, 25 117^2 float64) and five 11-pass solves (the GEMV chain of the
triangular solves).  Prints one JSON line with the vertex count."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import kernels, synthetic  # noqa: E402

device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
model = sc.factorize_model(device=device, current_units="uA")
n = len(device.meshes["washer0"].sites)
fd = model.film_data["washer0"]
C = torch.from_numpy(device.meshes["washer0"].operators.C).cuda()
ld = kernels.padded_ld(n, "float64")
Q = torch.empty((n, ld), dtype=torch.float64, device="cuda")
for _ in range(5):
    kernels.q_assemble(fd.xy, fd.w, C, "float64", out=Q, ld=ld)
for _ in range(5):
    sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=10)
torch.cuda.synchronize()
print(json.dumps({"vertices_per_film": n}))
