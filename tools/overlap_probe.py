"""Do the HBM-bound triangular solves of one film and the VALU-bound coupling sum of the other overlap?  (development aid)

In a two-film Jacobi iteration the passes form TWO independent chains (solve.py:491-536: every film sees the previous
iterate of the other):  sA(p) -> cAB(p) -> sB(p+1) -> cBA(p+1) -> sA(p+2) ...  and the same starting with sB(0).
Times 11 passes of config H's two solve + coupling kernels (a) as the solver issues them today -- both films' solves in
one batch, then both couplings, one stream -- and (b) as two chains on two streams, the second started half a period
late, so that one chain's solve (HBM) runs beside the other's coupling sum (FP64 vector ALU).

    python tools/overlap_probe.py [K=91]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import superscreen_amd as sc  # noqa: E402
from superscreen_amd import kernels, synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float64")
model = sc.factorize_model(device=device, current_units="uA")
films = list(device.films)
fd = {f: model.film_data[f] for f in films}
sysm = {f: model.film_systems[f] for f in films}
P = 11
rhs = {f: torch.randn(kernels.chol_padded_n(sysm[f].chol.n), dtype=torch.float64, device="cuda") for f in films}
for f in films:
    rhs[f][sysm[f].chol.n:] = 0
J = {f: torch.randn(fd[f].n, 2, dtype=torch.float64, device="cuda") for f in films}
rows = {f: sysm[f].indices_device for f in films}
xy_rows = {f: fd[f].xy.index_select(0, rows[f]).contiguous() for f in films}
out = {f: torch.empty(rows[f].numel(), dtype=torch.float64, device="cuda") for f in films}
A, B = films


def solve(f):
    kernels.chol_solve_batch([sysm[f].chol], [rhs[f]], padded=True)


def couple(src, tgt):
    s = fd[src]
    kernels.biot_savart(s.xy, s.w_t, J[src], xy_rows[tgt], 0.5, out[tgt], accumulate=False,
                        src_begin=s.src_range[0], src_end=s.src_range[1])


def timed(fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[1:]))


def today():
    for p in range(P):
        kernels.chol_solve_batch([sysm[A].chol, sysm[B].chol], [rhs[A], rhs[B]], padded=True)
        couple(B, A)
        couple(A, B)


def solves_only():
    for p in range(P):
        kernels.chol_solve_batch([sysm[A].chol, sysm[B].chol], [rhs[A], rhs[B]], padded=True)


def couplings_only():
    for p in range(P):
        couple(B, A)
        couple(A, B)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_chains(shift=True):
    main = torch.cuda.current_stream()
    start = torch.cuda.Event()
    start.record(main)
    first = torch.cuda.Event()
    with torch.cuda.stream(s1):
        s1.wait_event(start)
        for p in range(P):
            f, g = (A, B) if p % 2 == 0 else (B, A)
            solve(f)
            if p == 0:
                first.record(s1)
            couple(f, g)
    with torch.cuda.stream(s2):
        s2.wait_event(first if shift else start)
        for p in range(P):
            f, g = (B, A) if p % 2 == 0 else (A, B)
            solve(f)
            couple(f, g)
    main.wait_stream(s1)
    main.wait_stream(s2)


print(f"K={K}: unknowns {[sysm[f].chol.n for f in films]}, {P} passes")
print(f"  as today (batched solves, then couplings, one stream)  {timed(today):7.2f} ms")
print(f"  solves alone                                           {timed(solves_only):7.2f} ms")
print(f"  couplings alone                                        {timed(couplings_only):7.2f} ms")
print(f"  two chains on two streams, second half a period late   {timed(lambda: two_chains(True)):7.2f} ms")
print(f"  two chains on two streams, started together            {timed(lambda: two_chains(False)):7.2f} ms")
