"""Q assembly: the one-shot pieces against the row strips (SSA_Q_FORM=strips), same box, same buffers; and the two
forms' results against each other (off-diagonal entries: the same arithmetic, bit for bit; diagonal: the row sums
are added in another order, equal to rounding).   python tools/q_form_timing.py [K ...]   (development aid)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superscreen_amd import kernels, synthetic  # noqa: E402
from superscreen_amd.mesh import Mesh  # noqa: E402

Ks = [int(a) for a in sys.argv[1:]] or [91, 129]
for K in Ks:
    sites, elements, _ = synthetic.ring_disk_mesh(K)
    mesh = Mesh.from_triangulation(sites, elements)
    ops = mesh.operators
    xy, w, C = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (sites, ops.weights, ops.C))
    n = len(sites)
    for dtype in ("float64", "float32"):
        s = 8 if dtype == "float64" else 4
        ld = kernels.padded_ld(n, dtype)
        Q = torch.empty((n, ld), dtype=torch.float64 if s == 8 else torch.float32, device="cuda")
        res = {}
        for form in ("strips", "oneshot", "strips", "oneshot"):
            os.environ["SSA_Q_FORM"] = form
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for rep in range(6):
                e0.record()
                for _ in range(4):
                    _, qd = kernels.q_assemble(xy, w, C, dtype, out=Q, ld=ld)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4)
            ms = float(np.median(ts[1:]))
            print(f"n = {n} {dtype} {form:8s}: {ms:7.3f} ms  {n * n * s / ms / 1e9:6.2f} TB/s = {n * n * s / ms / 1e9 / 8:.3f} of 8 TB/s",
                  flush=True)
            res[form] = (Q[:, :n].clone(), qd.clone())
        a, b = res["strips"], res["oneshot"]
        off = ~torch.eye(n, dtype=torch.bool, device="cuda")
        print(f"   off-diagonal entries identical: {bool(torch.equal(a[0][off], b[0][off]))};  diagonal max rel diff "
              f"{float(((a[1] - b[1]).abs() / a[1].abs()).max()):.2e}", flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            Q.zero_()
        e1.record()
        torch.cuda.synchronize()
        print(f"   fill of the same buffer: {n * ld * s / (e0.elapsed_time(e1) / 4) / 1e9:.2f} TB/s", flush=True)
        del Q, res, a, b
    os.environ.pop("SSA_Q_FORM", None)
