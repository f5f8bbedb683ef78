"""The coupling sum of one ordered film pair of config H (solver/solve.py:28-73) by the vector-ALU kernel
(ssa_biot_savart: what solve() runs per pass) and by the MFMA pair kernels with ONE vector (ssa_biot_savart_multi:
what solve_sweep runs): time and difference (development aid).    python tools/coupling_probe.py [K=91]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import kernels, synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")
model = sc.factorize_model(device=device, current_units="uA")
A, B = list(device.films)
s, t = model.film_data[A], model.film_data[B]
rows = model.film_systems[B].indices_device
xy_rows = t.xy.index_select(0, rows).contiguous()
g = torch.Generator(device="cuda").manual_seed(0)
J = torch.randn(s.n, 2, dtype=torch.float64, device="cuda", generator=g)
out1 = torch.empty(rows.numel(), dtype=torch.float64, device="cuda")
out2 = torch.zeros((rows.numel(), 1), dtype=torch.float64, device="cuda")
b, e = s.src_range
xy_src, w_src = s.xy[b:e].contiguous(), s.w_t[b:e].contiguous()
J1 = J[b:e].reshape(e - b, 1, 2).contiguous()


def valu():
    kernels.biot_savart(s.xy, s.w_t, J, xy_rows, 0.5, out1, accumulate=False, src_begin=s.src_range[0], src_end=s.src_range[1])


def mfma():
    kernels.biot_savart_multi(xy_src, w_src, J1, xy_rows, 0.5, out2, accumulate=False)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


pairs = rows.numel() * (s.src_range[1] - s.src_range[0])
tv, tm = timed(valu), timed(mfma)
print(f"K={K}: {rows.numel()} targets x {s.src_range[1] - s.src_range[0]} sources = {pairs / 1e6:.0f} Mpairs")
print(f"  vector-ALU kernel        {tv:7.1f} us   {pairs / tv / 1e6:.2f} Tpair/s")
print(f"  MFMA pair kernel, 1 vec  {tm:7.1f} us   {pairs / tm / 1e6:.2f} Tpair/s")
print(f"  max rel difference {float((out1 - out2[:, 0]).abs().max() / out1.abs().max()):.2e}")
