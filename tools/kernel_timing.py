"""Ad-hoc per-kernel timing on one MI355X (development aid; bench.py is the contract).

    python tools/kernel_timing.py [--n 50311] [--lu 16384] [--what q,gemm,lu,solve,gemv,bs,fill]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superscreen_amd import kernels as K  # noqa: E402
from superscreen_amd import synthetic  # noqa: E402


def timeit(fn, reps=5, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=129)
    ap.add_argument("--luK", type=int, default=91)
    ap.add_argument("--what", default="fill,q,a,gemm,lu,solve,gemv,bs")
    ap.add_argument("--dtype", default="float64")
    args = ap.parse_args()
    what = set(args.what.split(","))
    dt = getattr(torch, args.dtype)
    es = 8 if args.dtype == "float64" else 4
    print("device:", K.device_info())

    if "fill" in what:
        buf = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
        med, mn = timeit(lambda: K.fill_probe(buf), reps=5)
        print(f"fill 8 GiB: {med*1e3:.3f} ms  -> {buf.numel()/med/1e12:.2f} TB/s (best {buf.numel()/mn/1e12:.2f})")
        del buf

    sites, elements, dr = synthetic.ring_disk_mesh(args.K)
    n = len(sites)
    from superscreen_amd import fem
    from superscreen_amd.mesh import MeshOperators

    w = fem.vertex_areas(sites, elements)
    C = MeshOperators.C_vector(sites)
    xy_d, w_d, C_d = (torch.from_numpy(a).cuda() for a in (sites, w, C))
    print(f"mesh K={args.K} n={n}")

    if "q" in what:
        ld = K.padded_ld(n, args.dtype)
        Q = torch.empty((n, ld), dtype=dt, device="cuda")
        med, mn = timeit(lambda: K.q_assemble(xy_d, w_d, C_d, args.dtype, out=Q, ld=ld), reps=5)
        print(f"q_assemble n={n}: {med*1e3:.3f} ms -> {n*n*es/med/1e12:.3f} TB/s algorithmic (best {n*n*es/mn/1e12:.3f})")
        med, mn = timeit(lambda: K.q_assemble(xy_d, w_d, C_d, args.dtype, want_Q=False), reps=5)
        print(f"q_diag only n={n}: {med*1e3:.3f} ms -> {n*n/med/1e9:.1f} Gpair/s")
        if "gemv" in what:
            g = torch.randn(n, dtype=dt, device="cuda")
            wd = w_d.to(dt)
            med, mn = timeit(lambda: K.gemv(Q, n, n, g, xscale=wd), reps=5)
            print(f"gemv Q@(w g) n={n}: {med*1e3:.3f} ms -> {n*n*es/med/1e12:.3f} TB/s")
            _, qd = K.q_assemble(xy_d, w_d, C_d, args.dtype, want_Q=False)
            med, mn = timeit(lambda: K.self_field(xy_d, w_d, qd, g), reps=5)
            print(f"self_field matrix-free n={n}: {med*1e3:.3f} ms -> {n*n/med/1e9:.1f} Gpair/s")
        del Q

    if "bs" in what:
        J = torch.randn((n, 2), dtype=torch.float64, device="cuda")
        out = torch.zeros(n, dtype=dt, device="cuda")
        med, mn = timeit(lambda: K.biot_savart(xy_d, w_d.to(dt), J, xy_d, 0.5, out, accumulate=False), reps=5)
        print(f"biot_savart n={n}: {med*1e3:.3f} ms -> {n*n/med/1e9:.1f} Gpair/s = {16*n*n/med/1e12:.2f} TFLOP/s (16 flop/pair)")

    if "gemm" in what:
        for (M, N, Kd) in [(8192, 8192, 256), (16384, 16384, 256), (32768, 32768, 256), (8192, 8192, 8192), (32768, 192, 64)]:
            A = torch.randn((M, Kd), dtype=dt, device="cuda")
            B = torch.randn((Kd, N), dtype=dt, device="cuda")
            Cm = torch.randn((M, N), dtype=dt, device="cuda")
            med, mn = timeit(lambda: K.gemm(A, B, Cm, M, N, Kd, alpha=-1.0, beta=1.0), reps=5)
            print(f"gemm {M}x{N}x{Kd}: {med*1e3:.3f} ms -> {2*M*N*Kd/med/1e12:.2f} TFLOP/s (best {2*M*N*Kd/mn/1e12:.2f})")
            del A, B, Cm

    if "lu" in what or "solve" in what:
        from matplotlib.path import Path

        for Klu in sorted({26, 58, args.luK}):
            # representative matrix: -A of a disk film on the K-ring mesh (diagonally dominant)
            s2, e2, dr2 = synthetic.ring_disk_mesh(Klu)
            Kf = synthetic.film_rings(Klu)
            w2 = fem.vertex_areas(s2, e2)
            lap = fem.laplace_operator(s2, e2, w2).tocsr()
            lap.sort_indices()
            inside = Path(synthetic.circle_points((Kf + 0.5) * dr2), closed=True).contains_points(s2)
            ix = np.setdiff1d(np.where(inside)[0], fem.boundary_indices(e2)).astype(np.int64)
            nlu = len(ix)
            put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            xy2, wd2, C2 = put(s2), put(w2), put(MeshOperators.C_vector(s2))
            _, qd2 = K.q_assemble(xy2, wd2, C2, args.dtype, want_Q=False)
            lap_d = (put(lap.indptr.astype(np.int64)), put(lap.indices.astype(np.int64)), put(lap.data))
            Lam = torch.full((len(s2),), 0.1, dtype=torch.float64, device="cuda")
            ixd = put(ix)
            asm = lambda: K.system_assemble(xy2, wd2, qd2, Lam, *lap_d, ixd, ixd, sign=-1.0, dtype=args.dtype)
            med, mn = timeit(asm, reps=3)
            print(f"system_assemble n_i={nlu}: {med*1e3:.3f} ms -> {nlu*nlu*es/med/1e12:.3f} TB/s")
            A0 = asm()
            A = A0.clone()
            ts = []
            for _ in range(3):
                A.copy_(A0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                f = K.lu_factor(A, nlu)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t2 = min(ts)
            print(f"lu_factor n={nlu}: {[round(t*1e3,1) for t in ts]} ms -> {2/3*nlu**3/t2/1e12:.2f} TFLOP/s; info={f.info}; "
                  f"pivots off-diagonal: {int((f.ipiv.cpu() != torch.arange(nlu, dtype=torch.int32)).sum())}")
            b = torch.randn(nlu, dtype=dt, device="cuda")
            x = K.lu_solve(f, b)
            r = (A0[:, :nlu] @ x - b).abs().max().item() / b.abs().max().item()
            med, mn = timeit(lambda: K.lu_solve_permuted(f, b.clone()), reps=5)
            print(f"lu_solve n={nlu} nrhs=1: {med*1e3:.3f} ms -> {nlu*nlu*es/med/1e12:.3f} TB/s; residual {r:.2e}")
            b64 = torch.randn((nlu, 64), dtype=dt, device="cuda")
            med, mn = timeit(lambda: K.lu_solve_permuted(f, b64.clone()), reps=3)
            print(f"lu_solve n={nlu} nrhs=64: {med*1e3:.3f} ms -> {2*nlu*nlu*64/med/1e12:.2f} TFLOP/s")
            del A, A0, f


if __name__ == "__main__":
    main()
