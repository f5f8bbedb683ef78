"""Headline benchmark: self-consistent solves/sec (+ Q-assembly GB/s) on the 50k-vertex
two-film device of BASELINE.json, on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload ("config H" of SURVEY.md section 8d, the configuration the metric is quoted on):
washer (z = 0) + shield disk (z = 0.5 um), both on the K = 91 synthetic ring mesh: 25 117
vertices per film = 50 234 vertices, 20 419 / 18 150 unknowns, Lambda = 0.1 um, uniform
applied field, float64.

One STEP = one cold self-consistent solve of that device: for both films regenerate the kernel
diagonal, assemble the film system (fused Q w - Lambda Del2 tiles; as the symmetric positive
definite diag(w) A) and factor it (Cholesky on MFMA; LU fallback), then 1 + ITER passes of
solve_film over both films and ITER rounds of inter-film Biot-Savart coupling (Jacobi, ITER =
10 as in SURVEY config 3), including the per-iteration Solution objects copied to the host.
Mesh geometry and sparse operators are resident in HBM before the timed region starts.

N > 1: independent applied-field values are sharded over the ranks (no data-path collective;
weak scaling: every rank runs K steps of the same size); value = total solves / max-over-ranks
time.

The JSON line also carries
  roofline      -- the dominant kernel: the MFMA trailing update of the factorization
                   (gemm_op_kernel<double, 0, 1, true>, the lower-tile SYRK of the Cholesky; the NN
                   gemm_kernel if the LU fallback ran; MFMA bound):
                   achieved = sum(algorithmic flops, K M (M + 1) per launch) / sum(kernel time),
                   both measured live with HIP events on the kernel's stream inside the library
                   (ssa_profile_*) over the timed region; `traffic` = memory-side L2 bytes per
                   launch from the rocprofv3 PMC passes of this same command
                   (profiles/r01_v7_syrk_pmc.json, corrections in tools/summarize_pmc.py), next to
                   the algorithmic bytes per launch (C tiles read + written, panel read once)
  cpu_baseline  -- the CPU oracle (numpy/scipy + OpenMP C ports of the numba kernels) timed on
                   this box's host cores on a bounded sample, rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (vendor nominal)
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec


def syrk_algorithmic_bytes(unknowns, elem=8):
    """Average algorithmic bytes of one trailing-update launch of the Cholesky schedule
    (chol.hip potrf_batch): per launch the lower 128 x 128 tiles of the trailing block behind the next
    panel are read and written once and the pending panels below them (256 columns, or 512 when the
    previous step's update was kept pending: large trailing matrices, every other step) are read once."""
    total, launches = 0.0, 0
    for n in unknowns:
        npad = -(-n // 256) * 256
        pend0 = 0
        for k0 in range(0, npad - 256, 256):
            right = npad - k0 - 256
            nw = min(right, 256)
            kp = k0 + 256 - pend0
            delay = kp < 512 and right > 8192 and ((k0 + npad) // 256) % 2 != 1
            if right > nw and not delay:
                m = right - nw
                nt = m // 128
                total += 2.0 * (nt * (nt + 1) // 2) * 128 * 128 * elem + m * kp * elem
                launches += 1
            if not delay:
                pend0 = k0 + 256
    return total / max(1, launches), launches


def pmc_traffic():
    """Memory-side bytes per SYRK launch from the committed rocprofv3 PMC summary (None if absent)."""
    path = os.path.join(ROOT, "profiles", "r01_v7_syrk_pmc.json")
    try:
        with open(path) as f:
            return float(json.load(f)["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def pmc_mfma():
    """MFMA-pipe busy fraction and effective clock of the SYRK launches from the committed rocprofv3 PMC
    summary (profiles/r01_v7_syrk_mfma_pmc.json; None if absent)."""
    path = os.path.join(ROOT, "profiles", "r01_v7_syrk_mfma_pmc.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return {"mfma_busy_fraction_of_active_cycles": float(d["mfma_busy_fraction_of_active_cycles"]),
                "effective_clock_GHz": float(d["effective_clock_GHz"]),
                "source": "profiles/r01_v7_syrk_mfma_pmc.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                          "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 pass of this command)"}
    except (OSError, KeyError, ValueError):
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--K", type=int, default=91, help="rings of the synthetic mesh (91 -> 25 117 vertices/film)")
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--cpu-sample-K", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(sample_K: int, target_K: int, iterations: int):
    """Times the CPU oracle on a K = sample_K two-film device and extrapolates every phase to
    the benchmark size with its complexity law (Q, A, solve, coupling ~ n^2; LU ~ n_i^3)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_oracle
    import cpu_kernels
    import scipy.linalg as la
    import superscreen_oracle as orc
    from matplotlib.path import Path

    from superscreen_amd import synthetic

    build_oracle.build(verbose=False)

    def sizes(K):
        Kf = synthetic.film_rings(K)
        n = synthetic.num_vertices(K)
        ni_disk = 1 + 3 * Kf * (Kf + 1)
        Kh = Kf // 3
        return n, ni_disk, ni_disk - (1 + 3 * Kh * (Kh + 1))

    sites, elements, dr = synthetic.ring_disk_mesh(sample_K)
    Kf = synthetic.film_rings(sample_K)
    mesh = orc.make_mesh(sites, elements, build_Q=False)      # sparse operators: set-up, untimed
    in_film = Path(synthetic.circle_points((Kf + 0.5) * dr), closed=True).contains_points(sites)
    in_hole = Path(synthetic.circle_points((Kf // 3 + 0.5) * dr, 201), closed=True).contains_points(sites)
    C = orc.C_vector(sites)
    t = {}
    t0 = time.perf_counter()
    q = cpu_kernels.q_matrix(sites)                            # distance.py:87 (OpenMP port)
    diag = -(C + np.einsum("ij, j -> i", q, mesh.weights)) / mesh.weights
    np.fill_diagonal(q, diag)
    mesh.Q = -q                                                # device/mesh.py:453-458
    t["q_assembly"] = time.perf_counter() - t0
    films = []
    t["a_assembly"] = t["lu"] = 0.0
    for name, holes, z0 in (("washer", {"hole": in_hole}, 0.0), ("disk", {}, 0.5)):
        t0 = time.perf_counter()
        f = orc.make_film(name, mesh, z0=z0, Lambda=0.1, in_film=in_film, holes_mask=holes, factorize=False)
        t["a_assembly"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        f.lu_piv = la.lu_factor(-f.A)                          # solver/solve_film.py:279
        t["lu"] += time.perf_counter() - t0
        films.append(f)
    conv = orc.field_conversion_mT_to_uA_per_um()
    applied = {f.name: conv * np.ones(len(sites)) for f in films}
    t0 = time.perf_counter()
    sols = {f.name: orc.solve_film(f, applied[f.name], field_conversion=conv) for f in films}
    t_pass = time.perf_counter() - t0
    t0 = time.perf_counter()
    for src, tgt in ((films[0], films[1]), (films[1], films[0])):
        cpu_kernels.biot_savart_film_to_film(
            film1_sites=sites, film1_z0=src.z0, film1_areas=src.weights,
            film1_J=sols[src.name].current_density, film2_sites=sites, film2_z0=tgt.z0)
    t_cpl = time.perf_counter() - t0
    ns, nis_d, nis_w = sizes(sample_K)
    nt, nit_d, nit_w = sizes(target_K)
    r2 = (nt / ns) ** 2
    ri2 = (nit_d ** 2 + nit_w ** 2) / (nis_d ** 2 + nis_w ** 2)
    ri3 = (nit_d ** 3 + nit_w ** 3) / (nis_d ** 3 + nis_w ** 3)
    # one q_matrix per mesh; both films share the mesh in the sample, the device has 2 meshes
    est = (2 * t["q_assembly"] * r2 + t["a_assembly"] * ri2 + t["lu"] * ri3
           + (iterations + 1) * t_pass * (ri2 + r2) / 2 + iterations * t_cpl * r2)
    return {
        "value": 1.0 / est,
        "unit": "solves/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": (f"oracle (numpy/scipy LAPACK + OpenMP C ports of the numba kernels) on the K={sample_K} "
                   f"two-film device (n={ns}/film, n_i={nis_w}+{nis_d}); phases extrapolated to K={target_K} "
                   f"(n={nt}, n_i={nit_w}+{nit_d}) with n^2 (Q, A, solve, coupling) and n_i^3 (LU) laws"),
        "sample_seconds": {"q_assembly_x1": t["q_assembly"], "a_assembly": t["a_assembly"], "lu": t["lu"],
                           "solve_pass": t_pass, "coupling_round": t_cpl},
        "estimated_seconds_per_solve": est,
    }


def main():
    args = parse_args()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path.")
    # BENCH_SHARE_GPU=1 (testing aid for single-GPU boxes): all ranks use cuda:0 and gloo carries the
    # barrier / max-reduce, so that the N > 1 control flow can be exercised without N GPUs
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import superscreen_amd as sc
    from superscreen_amd import _hip, kernels, synthetic

    lib = _hip.load_library()
    device = synthetic.make_stack_device(args.K, ("washer", "disk"), solve_dtype="float64")
    n = len(device.meshes["washer0"].sites)

    def step(i):
        field = 0.1 * (1 + rank + world * i)  # mT; every rank / step solves a different field
        model = sc.factorize_model(device=device, current_units="uA")
        sols = sc.solve(model=model, applied_field=sc.ConstantField(field), field_units="mT",
                        iterations=args.iterations, progress_bar=False)
        return model, sols

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Every step drops the previous step's model before it builds its own (as a loop over devices or
    # a parameter scan does), in the warm-up exactly as in the timed region: the timed steps then reuse
    # the HBM blocks the warm-up left in the caching allocator instead of paying a one-time 7 GB
    # hipMalloc for a second live model inside the timed region (150-270 ms on a fresh box).
    model = sols = None
    for i in range(args.warmup):
        model = sols = None
        model, sols = step(-1 - i)
    barrier()
    _hip.check(lib.ssa_profile_begin(), "ssa_profile_begin")
    t0 = time.perf_counter()
    for i in range(args.steps):
        model = sols = None
        model, sols = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = {}
    for kind, label in ((0, "ssa::gemm_kernel<double, true> (NN: LU trailing / in-panel updates)"),
                        (1, "ssa::gemm_op_kernel<double, 0, 1, true> (SYRK on the lower tiles: Cholesky trailing update)")):
        ms, fl, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        _hip.check(lib.ssa_profile_read(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt)),
                   "ssa_profile_read")
        prof[label] = (ms.value, fl.value, cnt.value)
    _hip.check(lib.ssa_profile_end(), "ssa_profile_end")
    dom_label = max(prof, key=lambda k: prof[k][0])
    gemm_ms, gemm_fl, gemm_n = (ctypes.c_double(prof[dom_label][0]), ctypes.c_double(prof[dom_label][1]),
                                ctypes.c_int64(prof[dom_label][2]))
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    total_solves = args.steps * world
    value = total_solves / elapsed

    if rank == 0:
        # sanity: the last step's answer is a converged, finite screening solution
        g = sols[-1].film_solutions["disk1"].stream
        assert np.isfinite(g).all() and len(sols) == args.iterations + 1
        extras = {}
        # Q-assembly throughput of one film's dense kernel matrix (not part of the timed step)
        fd = model.film_data["washer0"]
        ld = kernels.padded_ld(n, "float64")
        Q = torch.empty((n, ld), dtype=torch.float64, device="cuda")
        C = torch.from_numpy(device.meshes["washer0"].operators.C).cuda()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kernels.q_assemble(fd.xy, fd.w, C, "float64", out=Q, ld=ld)
        ts = []
        for _ in range(5):
            e0.record()
            kernels.q_assemble(fd.xy, fd.w, C, "float64", out=Q, ld=ld)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        tq = float(np.median(ts))
        extras["q_assembly_GBps"] = n * n * 8 / tq / 1e9
        extras["q_assembly_ms"] = tq * 1e3
        extras["q_assembly_frac_of_hbm_peak"] = extras["q_assembly_GBps"] / HBM_PEAK_GBPS
        del Q
        # what a bare register-only v_mfma_f64 stream (8 waves per SIMD, no memory, no LDS) is held at on
        # this box right now: 45-62 TFLOP/s observed -- the chip lowers its clock under a pure MFMA load,
        # so this is a power-management reading, not a ceiling (the SYRK engine itself reaches 67 TFLOP/s
        # at K = 2048, tools/probes/syrk_k_probe.py); the roofline fraction is priced against 78.6
        extras["fp64_mfma_register_only_TFLOPs"] = kernels.mfma_probe(4000)
        # field map above the device: 512 x 512 image, all-pairs Biot-Savart of one film
        gx = torch.linspace(-6.0, 6.0, 512, dtype=torch.float64, device="cuda")
        ev = torch.stack([gx.repeat_interleave(512), gx.repeat(512), torch.full((512 * 512,), 1.0, dtype=torch.float64,
                                                                                device="cuda")], dim=1).contiguous()
        Jd = torch.from_numpy(sols[-1].film_solutions["washer0"].current_density).cuda()
        kernels.sheet_field(fd.xy, fd.w, Jd, 0.0, ev, 1e-7, True)
        e0.record()
        kernels.sheet_field(fd.xy, fd.w, Jd, 0.0, ev, 1e-7, True)
        e1.record()
        torch.cuda.synchronize()
        extras["field_map_512x512_vector_ms"] = e0.elapsed_time(e1)
        extras["field_map_Tpairs_per_s"] = 512 * 512 * n / (e0.elapsed_time(e1) * 1e-3) / 1e12
        # warm (pre-factorized) self-consistent solves
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(3):
            sc.solve(model=model, applied_field=sc.ConstantField(0.3 + i), iterations=args.iterations)
        torch.cuda.synchronize()
        extras["warm_self_consistent_solves_per_s"] = 3 / (time.perf_counter() - t1)
        # BASELINE config 4: a 64-value applied-field scan of the same device carried as the columns of
        # one multi-right-hand-side solve (solve_sweep; final iterate of every field returned as
        # Solutions on the host).  Reported per GPU; scans shard across ranks without a collective.
        scan = [0.05 * (k + 1) for k in range(64)]
        sc.solve_sweep(model, scan[:8], iterations=args.iterations, all_iterations=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        swept = sc.solve_sweep(model, scan, iterations=args.iterations, all_iterations=False)
        torch.cuda.synchronize()
        extras["field_sweep_64_values_solves_per_s"] = len(swept) / (time.perf_counter() - t1)
        del swept
        # ... and 8 values: one rank's share when the 64-value scan is sharded over 8 GPUs
        t1 = time.perf_counter()
        swept = sc.solve_sweep(model, scan[:8], iterations=args.iterations, all_iterations=False)
        torch.cuda.synchronize()
        extras["field_sweep_8_values_solves_per_s"] = len(swept) / (time.perf_counter() - t1)
        del swept
        # iterations needed for max|dg|/max|g| < 1e-8 (the reference has no convergence test)
        conv = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=200, tolerance=1e-8,
                        return_solutions=True)
        extras["iterations_to_1e-8"] = len(conv) - 1
        achieved = gemm_fl.value / (gemm_ms.value * 1e-3) / 1e12 if gemm_ms.value > 0 else 0.0
        out = {
            "metric": "self_consistent_solves_per_sec",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"config H: cold self-consistent solve of the 2-film washer+disk device, K={args.K} ring "
                             f"mesh, {n} vertices/film ({2 * n} total), Lambda=0.1 um, uniform field, "
                             f"{args.iterations} Jacobi iterations, factorization included in every step"),
                "vertices_per_film": n,
                "unknowns": [int(len(s.indices)) for s in model.film_systems.values()],
                "iterations": args.iterations,
                "parallelism": f"field-sweep sharding x{world} (no data-path collective)",
            },
            "roofline": {
                "kernel": dom_label + ", v_mfma_f64_16x16x4_f64",
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "traffic": pmc_traffic() if "gemm_op_kernel" in dom_label else None,
                "traffic_source": "profiles/r01_v7_syrk_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  "this command; bytes per launch, fetch x2 per MI355X_MICROARCH.md)",
                "algorithmic_bytes_per_launch": syrk_algorithmic_bytes(
                    [int(len(s.indices)) for s in model.film_systems.values()])[0],
                "launches": int(gemm_n.value),
                "avg_launch_us": gemm_ms.value * 1e3 / max(1, gemm_n.value),
                "avg_launch_gflop": gemm_fl.value / max(1, gemm_n.value) / 1e9,
                "mfma_pipe": pmc_mfma() if "gemm_op_kernel" in dom_label else None,
            },
            "extras": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_K, args.K, args.iterations)
            out["extras"]["gpu_over_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
