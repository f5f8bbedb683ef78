"""Headline benchmark: self-consistent solves/sec (+ Q-assembly GB/s) on the 50k-vertex
two-film device of BASELINE.json, on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload ("config H" of SURVEY.md section 8d, the configuration the metric is quoted on):
washer (z = 0) + shield disk (z = 0.5 um), both on the K = 91 synthetic ring mesh: 25 117
vertices per film = 50 234 vertices, 18 150 / 20 419 unknowns, Lambda = 0.1 um, uniform
applied field, float64.

One STEP = one cold self-consistent solve of that device: for both films regenerate the kernel
diagonal, assemble the film system (fused Q w - Lambda Del2 tiles; as the symmetric positive
definite diag(w) A) and factor it (Cholesky on MFMA; LU fallback), then 1 + ITER passes of
solve_film over both films and ITER rounds of inter-film Biot-Savart coupling (Jacobi, ITER =
10 as in SURVEY config 3), including the per-iteration Solution objects copied to the host.
Mesh geometry and sparse operators are resident in HBM before the timed region starts.

The headline line is the same at every N: at N > 1 every rank runs K such steps on its own applied
field (independent solves: weak scaling, no data-path collective); value = total solves /
max-over-ranks time.  What the other BASELINE configurations do at N GPUs is reported in `extras`
(all ranks take part; RCCL = torch.distributed backend "nccl"):
  config4_*   64-value applied-field scan of the config-3 device (2 x 19 927 vertices) through
              parallel.solve_sweep_sharded: `strong` = 64 values over the N ranks, every rank's own
              factorization included; `weak` = 64 values per rank.  No collective in the data path.
  config5_*   cold solve of the 4-film stack (4 x 30 301 vertices, 10 Jacobi iterations) with the
              films placed on the ranks (parallel.FilmPlacement, N <= 4: one sum all-reduce of the O(n)
              result vectors per pass) or, when there are more ranks than films (N = 8), with every
              ordered film pair split by source slice (parallel.CouplingPlan: one fused all-reduce of the
              coupling vector per iteration).

The JSON line also carries
  roofline      -- the dominant kernel: the MFMA trailing update of the factorization
                   (gemm_op_kernel<double, 0, 1, true>, the lower-tile SYRK of the Cholesky; the NN
                   gemm_kernel if the LU fallback ran; MFMA bound):
                   achieved = sum(algorithmic flops, K M (M + 1) per launch) / sum(kernel time),
                   both measured live with HIP events on the kernel's stream inside the library
                   (ssa_profile_*) over the timed region; `traffic` = memory-side L2 bytes per
                   launch from the rocprofv3 PMC passes of this same command (profiles/, corrections in
                   tools/summarize_pmc.py), next to the algorithmic bytes per launch (C tiles read +
                   written, panel read once).  The PMC summaries record the average launch duration
                   and flops of the run they were collected in; if the live flops per launch have moved
                   more than 5 % (the schedule changed) or the duration more than 15 % (the kernel
                   changed; boxes differ by 12 %) away from them, `traffic_stale` is true.
  cpu_baseline  -- the CPU oracle (numpy/scipy + OpenMP C ports of the numba kernels) timed on this
                   box's host cores ON THE SAME K = 91 DEVICE (no extrapolation), thread count chosen by
                   a short LU sweep; rank 0 at N = 1 only.
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
# HIP multiplexes a process' streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues per priority level; the
# factorization schedules use two high-priority chain streams per film, so the 4-film stack of config 5 wants 8
# (DESIGN_HISTORY.md, round 4).  An application-level choice: set here, before the HIP runtime starts, not by the package.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (vendor nominal)
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec
PMC_TRAFFIC_FILES = ("r06_syrk_pmc.json", "r05_syrk_pmc.json", "r04_syrk_pmc.json", "r03_syrk_pmc.json", "r02b_syrk_pmc.json")
PMC_MFMA_FILES = ("r06_syrk_mfma_pmc.json", "r05_syrk_mfma_pmc.json", "r04_syrk_mfma_pmc.json", "r03_syrk_mfma_pmc.json", "r02b_syrk_mfma_pmc.json")


def chol_schedule(unknowns, elem=8):
    """The trailing-update launches of the Cholesky schedule (chol.hip potrf_batch) for films with the given
    numbers of unknowns: per launch the lower 128 x 128 tiles of the trailing block are read and written once and
    the pending panels below them (256 columns, or 512 when the previous step's update was kept pending: large
    trailing matrices, every other step) are read once.  The block starts behind the next panel.  Only the stream
    part of the schedule launches this kernel: once every film's trailing matrix is at most 10 240 columns the
    updates are tiles of the round launches.
    Returns (average algorithmic bytes per launch, launches per factorization)."""
    total, launches = 0.0, 0
    npads = [-(-n // 256) * 256 for n in unknowns]
    nmax = max(npads)
    upd0 = [0] * len(npads)
    for k0 in range(0, nmax - 256, 256):
        c = k0 + 256
        if nmax - c <= 10240:   # from here on the schedule runs as rounds (chol_tail_round_kernel, not this kernel)
            break
        for f, npad in enumerate(npads):
            if c >= npad:
                continue
            right = npad - c
            kp = c - upd0[f]
            delay = kp < 512 and right > 8192 and ((k0 + npad) // 256) % 2 != 1
            if right > 256 and not delay:
                m = npad - (c + 256)
                nt = m // 128
                total += 2.0 * (nt * (nt + 1) // 2) * 128 * 128 * elem + m * kp * elem
                launches += 1
            if not delay:
                upd0[f] = c
    return total / max(1, launches), launches


def load_profile(names):
    """First existing PMC summary of `names` under profiles/ -> (dict, relative path), else (None, None)."""
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                return json.load(f), "profiles/" + name
        except (OSError, ValueError):
            continue
    return None, None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--K", type=int, default=91, help="rings of the synthetic mesh (91 -> 25 117 vertices/film)")
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=60.0,
                    help="repeat the CPU phases (median) while the accumulated CPU time stays below this")
    ap.add_argument("--no-extras", action="store_true", help="headline line only (profiling runs)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on the same device, measured
# ---------------------------------------------------------------------------------------------------
def physical_cores():
    try:
        cores = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        return len(cores) or None
    except OSError:
        return None


def cpu_baseline(K: int, iterations: int, budget_s: float, gpu_solutions=None, gpu_field_mT=None):
    """One cold self-consistent solve of the K-ring two-film device by the CPU oracle, phase by phase, on
    this box's host cores: dense Q in float64 (OpenMP C port of the numba kernel, distance.py:87-115),
    A = Q[ix, ix] w - Lambda Del2 (solve_film.py:296-305), scipy lu_factor(-A) (:279), then 1 + iterations
    passes of solve_film (lu_solve, Q @ (w g), sparse gradients; :440-574) and `iterations` rounds of the
    all-pairs coupling (solve.py:28-73, OpenMP C port).  Nothing is extrapolated.

    With ``gpu_solutions`` (the Solutions of the last timed GPU step, applied field ``gpu_field_mT``) the oracle
    then runs that very solve -- the reference's ``gf = lu_solve(lu_piv, h)`` on ``lu_factor(-A)`` with
    ``A = Q[ix, ix] w - Lambda Del2`` (solve_film.py:296-305, 526-531) inside the Jacobi loop of solve.py:491-536,
    every iterate -- and the returned dict carries ``parity``: the stream-function max-rel-error
    ``max|g_gpu - g_ref| / max|g_ref|`` per film and iterate AT THE BENCHMARK SIZE (north_star: < 1e-6)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_oracle
    import cpu_kernels
    import scipy.linalg as la
    import superscreen_oracle as orc
    from matplotlib.path import Path
    from threadpoolctl import threadpool_limits

    from superscreen_amd import synthetic

    build_oracle.build(verbose=False)
    logical = os.cpu_count() or 1
    phys = physical_cores() or logical
    # thread sweep on a small LU: BLAS oversubscription (one thread per SMT sibling) can halve the rate
    rng = np.random.default_rng(0)
    n_probe = 6000
    probe = rng.standard_normal((n_probe, n_probe)) + n_probe * np.eye(n_probe)
    sweep = {}
    for t in sorted({max(1, phys // 16), max(1, phys // 8), max(1, phys // 4), max(1, 3 * phys // 8), max(1, phys // 2),
                     phys, logical}):
        with threadpool_limits(limits=t):
            la.lu_factor(probe[:1000, :1000])
            t0 = time.perf_counter()
            la.lu_factor(probe)
            sweep[t] = (2 / 3) * n_probe ** 3 / (time.perf_counter() - t0) / 1e9
    threads = max(sweep, key=sweep.get)
    del probe
    # ... and at the benchmark's own size (the LU of the larger film, n_i = 20 419 at K = 91): the small probe
    # collapses at high thread counts on some boxes where the large factorization does not.  The baseline uses the
    # thread count that is fastest AT SIZE.
    n_size = {91: 20419, 81: 16207, 129: 41419}.get(K, 0)
    sweep_at_size = {}
    if n_size and n_size <= 24000:
        big = rng.random((n_size, n_size))
        big[np.diag_indices(n_size)] += n_size
        for t in sorted({threads, min(32, phys), min(64, phys)}):
            if sweep_at_size and t > threads and sweep_at_size[max(sweep_at_size)] < 0.75 * max(sweep_at_size.values()):
                break   # the rate is falling with the thread count: a larger count would only cost its 15-20 s
            with threadpool_limits(limits=t):
                t0 = time.perf_counter()
                la.lu_factor(big, check_finite=False)
                sweep_at_size[t] = (2 / 3) * n_size ** 3 / (time.perf_counter() - t0) / 1e9
        threads = max(sweep_at_size, key=sweep_at_size.get)
        del big

    def median_timed(fn, spent, cap=3):
        """Median wall time of up to `cap` runs of fn, stopping early once the CPU budget is used up."""
        times, out = [], None
        while len(times) < cap and (not times or spent[0] + times[-1] < budget_s):
            t0 = time.perf_counter()
            out = fn()
            times.append(time.perf_counter() - t0)
            spent[0] += times[-1]
        return float(np.median(times)), len(times), out

    spent = [0.0]
    with threadpool_limits(limits=threads):
        sites, elements, dr = synthetic.ring_disk_mesh(K)
        Kf = synthetic.film_rings(K)
        mesh = orc.make_mesh(sites, elements, build_Q=False)      # sparse operators: set-up, untimed
        in_film = Path(synthetic.circle_points((Kf + 0.5) * dr), closed=True).contains_points(sites)
        in_hole = Path(synthetic.circle_points((Kf // 3 + 0.5) * dr, 201), closed=True).contains_points(sites)
        C = orc.C_vector(sites)
        n = len(sites)

        def build_Q():
            q = cpu_kernels.q_matrix(sites)                        # distance.py:87 (OpenMP port)
            diag = -(C + np.einsum("ij, j -> i", q, mesh.weights)) / mesh.weights
            np.fill_diagonal(q, diag)
            np.negative(q, out=q)                                  # device/mesh.py:453-458
            return q

        t_q, r_q, Q = median_timed(build_Q, spent)
        mesh.Q = Q
        films, t_a, t_lu, r_lu, lu_flops = [], 0.0, 0.0, [], 0.0
        for name, holes, z0 in (("washer", {"hole": in_hole}, 0.0), ("disk", {}, 0.5)):
            t0 = time.perf_counter()
            f = orc.make_film(name, mesh, z0=z0, Lambda=0.1, in_film=in_film, holes_mask=holes, factorize=False)
            t_a += time.perf_counter() - t0
            spent[0] += time.perf_counter() - t0

            def factor(f=f):
                return la.lu_factor(-f.A)                          # solver/solve_film.py:279

            t, r, f.lu_piv = median_timed(factor, spent)
            t_lu += t
            r_lu.append(r)
            lu_flops += (2 / 3) * len(f.film_indices) ** 3
            films.append(f)
        conv = orc.field_conversion_mT_to_uA_per_um()
        applied = {f.name: conv * np.ones(n) for f in films}
        sols = {}

        def one_pass():
            for f in films:
                sols[f.name] = orc.solve_film(f, applied[f.name], field_conversion=conv)

        t_pass, r_pass, _ = median_timed(one_pass, spent)

        def coupling_round():
            for src, tgt in ((films[0], films[1]), (films[1], films[0])):
                cpu_kernels.biot_savart_film_to_film(
                    film1_sites=sites, film1_z0=src.z0, film1_areas=src.weights,
                    film1_J=sols[src.name].current_density, film2_sites=sites, film2_z0=tgt.z0)

        t_cpl, r_cpl, _ = median_timed(coupling_round, spent)
        parity = None
        if gpu_solutions is not None:
            parity = oracle_parity(orc, cpu_kernels, films, gpu_field_mT, iterations, gpu_solutions)
        # ---- the reference's DEFAULT precision (solve_dtype="float32", device/device.py:57; BASELINE.md section 3 asks for
        # both): Q is still assembled in float64 and cast (solver/utils.py:290-292), A, sgetrf / sgetrs and Q @ (w g)
        # run in float32.  The float64 films are dropped first (their Q / A / LU are ~ 14 GB of host memory).
        f32 = None
        try:
            for f in films:
                f.A = f.lu_piv = None
            films32, t_a32, t_lu32, lu32_flops = [], 0.0, 0.0, 0.0
            for name, holes, z0 in (("washer", {"hole": in_hole}, 0.0), ("disk", {}, 0.5)):
                t0 = time.perf_counter()
                f = orc.make_film(name, mesh, z0=z0, Lambda=0.1, in_film=in_film, holes_mask=holes, factorize=False,
                                  dtype="float32")
                t_a32 += time.perf_counter() - t0
                t0 = time.perf_counter()
                f.lu_piv = la.lu_factor(-f.A, check_finite=False)      # sgetrf
                t_lu32 += time.perf_counter() - t0
                lu32_flops += (2 / 3) * len(f.film_indices) ** 3
                films32.append(f)
            applied32 = {f.name: (conv * np.ones(n)).astype(np.float32) for f in films32}
            t0 = time.perf_counter()
            for f in films32:
                orc.solve_film(f, applied32[f.name], field_conversion=conv)
            t_pass32 = time.perf_counter() - t0
            per_solve32 = 2 * t_q + t_a32 + t_lu32 + (iterations + 1) * t_pass32 + iterations * t_cpl
            f32 = {"value": 1.0 / per_solve32, "unit": "solves/s", "seconds_per_solve": per_solve32,
                   "sample_seconds": {"q_assembly_per_film_float64": t_q, "cast_and_a_assembly_both_films": t_a32,
                                      "lu_both_films": t_lu32, "solve_pass_both_films": t_pass32,
                                      "coupling_round_float64": t_cpl},
                   "lu_GFLOPs": lu32_flops / t_lu32 / 1e9, "threads_used": threads,
                   "note": "one run per phase; Q assembly and the coupling sums are float64 in the reference whatever the "
                           "solve_dtype (distance.py:101, solver/solve.py:508-515)"}
            del films32
        except MemoryError as exc:   # (a small host)
            f32 = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    # the device has one mesh per film (device.meshes, device/device.py): Q is built per film
    per_solve = 2 * t_q + t_a + t_lu + (iterations + 1) * t_pass + iterations * t_cpl
    return {
        "value": 1.0 / per_solve,
        "unit": "solves/s",
        "cores": threads,
        "kind": "port",
        "parity": parity,
        "measurement": "measured at the benchmark size, no extrapolation",
        "sample": (f"CPU oracle (numpy/scipy LAPACK + OpenMP C ports of the numba kernels) on the SAME K={K} two-film "
                   f"device (n={n}/film, n_i={len(films[0].film_indices)}+{len(films[1].film_indices)}), every phase "
                   f"timed at full size: seconds per solve = 2 x Q + A + LU(both films) + {iterations + 1} passes + "
                   f"{iterations} coupling rounds; median of up to 3 runs per phase within a {budget_s:.0f} s budget"),
        "sample_seconds": {"q_assembly_per_film": t_q, "a_assembly_both_films": t_a, "lu_both_films": t_lu,
                           "solve_pass_both_films": t_pass, "coupling_round": t_cpl},
        "repeats": {"q_assembly": r_q, "lu_per_film": r_lu, "solve_pass": r_pass, "coupling_round": r_cpl},
        "seconds_per_solve": per_solve,
        "lu_GFLOPs": lu_flops / t_lu / 1e9,
        "threads_used": threads,
        "logical_cpus": logical,
        "physical_cores": phys,
        "lu_thread_sweep_GFLOPs_n6000": {str(k): v for k, v in sweep.items()},
        "lu_GFLOPs_by_threads_at_size": {str(k): v for k, v in sweep_at_size.items()},
        # BASELINE.md section 3 says "all cores": this host's LAPACK is SLOWER on all of its cores than on the best
        # count (oversubscribed OpenBLAS threads), so the baseline above uses the best count and the all-cores rate is
        # printed beside it
        "all_cores": {"threads": phys, "lu_GFLOPs_n6000": sweep.get(phys), "best_threads_n6000": max(sweep, key=sweep.get),
                      "lu_GFLOPs_n6000_best_threads": max(sweep.values())},
        "float32": f32,
    }


def oracle_parity(orc, cpu_kernels, films, field_mT, iterations, gpu_solutions):
    """The oracle's own self-consistent solve of the benchmark device (first pass + `iterations` Jacobi rounds,
    solve.py:459-536; all sources use the previous iterate) against the GPU Solutions of the same solve."""
    t0 = time.perf_counter()
    trace = orc.solve(films, field_mT, iterations=iterations, biot_savart=cpu_kernels.biot_savart_film_to_film)
    gpu_names = list(gpu_solutions[0].film_solutions)            # device.films order = oracle film order
    assert len(gpu_names) == len(films) and len(gpu_solutions) == len(trace) == iterations + 1
    worst = {"stream": 0.0, "current_density": 0.0, "self_field": 0.0, "field_from_other_films": 0.0}
    per_film = {nm: 0.0 for nm in gpu_names}
    per_iterate = []

    def rel(a, b):
        return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / max(float(np.max(np.abs(b))), 1e-300))

    for it, sols in enumerate(trace):
        err_it = 0.0
        for f, nm in zip(films, gpu_names):
            ref, got = sols[f.name], gpu_solutions[it].film_solutions[nm]
            e = rel(got.stream, ref.stream)
            per_film[nm] = max(per_film[nm], e)
            err_it = max(err_it, e)
            worst["stream"] = max(worst["stream"], e)
            worst["current_density"] = max(worst["current_density"], rel(got.current_density, ref.current_density))
            worst["self_field"] = max(worst["self_field"], rel(got.self_field, ref.self_field))
            if it:
                worst["field_from_other_films"] = max(worst["field_from_other_films"],
                                                      rel(got.field_from_other_films, ref.field_from_other_films))
        per_iterate.append(err_it)
    fluxoid = fluxoid_parity(orc, films, trace, gpu_solutions)
    return {
        "max_rel_err_stream": worst["stream"],
        "max_rel_err_fluxoid": fluxoid["max_rel_err_fluxoid"],
        "fluxoid": fluxoid,
        "per_film": per_film,
        "per_iterate_stream": per_iterate,
        "max_rel_err_current_density": worst["current_density"],
        "max_rel_err_self_field": worst["self_field"],
        "max_rel_err_field_from_other_films": worst["field_from_other_films"],
        "iterations": iterations,
        "applied_field_mT": field_mT,
        "tolerance": 1e-6,
        "passed": bool(worst["stream"] < 1e-6 and fluxoid["max_rel_err_fluxoid"] < 1e-6),
        "definition": "max|g_gpu - g_ref| / max|g_ref| per film and iterate, all iterates of the last timed GPU step",
        "reference": ("CPU oracle: scipy lu_factor(-A) / lu_solve (solve_film.py:279, 526-531), A = Q[ix,ix] w - "
                      "Lambda Del2 (:296-305), Jacobi loop of solve.py:491-536, at the benchmark size"),
        "oracle_seconds": time.perf_counter() - t0,
    }


def fluxoid_parity(orc, films, trace, gpu_solutions):
    """Fluxoids of the benchmark solve against the oracle (north_star: "stream functions and fluxoids";
    solution.py:484-563, 565-609): a ring half way between the washer's hole and the film's rim, on every film (the
    washer's is the fluxoid of its hole, the disk's that of a simply connected region), every iterate.  Error of a
    fluxoid = the larger of |flux part - ref| and |supercurrent part - ref| over the larger of the two reference
    parts (both parts in mT um^2)."""
    device = gpu_solutions[0].device
    names = list(gpu_solutions[0].film_solutions)
    from superscreen_amd import Polygon, synthetic

    film_poly = device.films[names[0]].points
    r_film = float(np.max(np.hypot(film_poly[:, 0], film_poly[:, 1])))
    holes = list(device.holes.values())
    r_hole = float(np.max(np.hypot(holes[0].points[:, 0], holes[0].points[:, 1]))) if holes else 0.0
    ring = Polygon(points=synthetic.circle_points(0.5 * (r_hole + r_film), 301)).points
    worst, last = 0.0, {}
    for it, ref in enumerate(trace):
        for f, nm in zip(films, names):
            got = gpu_solutions[it].polygon_fluxoid(ring, film=nm, units="mT * um**2", with_units=False)
            want = orc.polygon_fluxoid_mT_um2(f, ref[f.name], ring, device.films[nm].points)
            scale = max(abs(want[0]), abs(want[1]))
            err = max(abs(got.flux_part - want[0]), abs(got.supercurrent_part - want[1])) / scale
            worst = max(worst, err)
            if it == len(trace) - 1:
                last[nm] = {"gpu_mT_um2": [got.flux_part, got.supercurrent_part], "oracle_mT_um2": list(want),
                            "fluxoid_Phi0": (got.flux_part + got.supercurrent_part) * 1e-15 / orc.PHI_0}
    return {"max_rel_err_fluxoid": worst, "ring_radius_um": 0.5 * (r_hole + r_film), "last_iterate": last,
            "definition": "max over films and iterates of max(|flux part - ref|, |supercurrent part - ref|) / "
                          "max(|ref flux part|, |ref supercurrent part|)"}


# ---------------------------------------------------------------------------------------------------
# BASELINE configs 2 and 3 (one GPU) and the other reading of the headline (2 x 50 311 vertices)
# ---------------------------------------------------------------------------------------------------
def config2_single_film(sc, torch, kernels):
    """Config 2: single-film 50 311-vertex disk (K = 129, 41 419 unknowns): (i) Q assembly alone (20.25 GB),
    (ii) the fused A assembly, (iii) the factorization, (iv) one solve_film."""
    from superscreen_amd import synthetic

    K2 = int(os.environ.get("BENCH_CONFIG2_K", "129"))            # testing aid: smaller meshes
    device = synthetic.make_stack_device(K2, ("disk",), solve_dtype="float64")
    name = list(device.films)[0]
    n = len(device.meshes[name].sites)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps=3, calls=4):
        # `calls` back-to-back calls between two events: the device-side rate (with one call per event pair the
        # device idles behind the first event until the host has issued the launch: 20-40 us on a 1-4 ms kernel)
        fn()
        ts = []
        for _ in range(reps):
            e0.record()
            for _ in range(calls):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / calls)
        return float(np.median(ts))

    model = sc.factorize_model(device=device, current_units="uA")            # warm-up: allocator, caches
    fd, system = model.film_data[name], model.film_systems[name]
    ni = len(system.indices)
    used_chol = system.chol is not None
    res = {"config2_vertices": n, "config2_unknowns": ni}
    ld = kernels.padded_ld(n, "float64")
    Q = torch.empty((n, ld), dtype=torch.float64, device="cuda")
    C = torch.from_numpy(device.meshes[name].operators.C).cuda()
    t = timed(lambda: kernels.q_assemble(fd.xy, fd.w, C, "float64", out=Q, ld=ld))
    res["config2_q_assembly_ms"] = t
    res["config2_q_assembly_GBps"] = n * n * 8 / (t * 1e-3) / 1e9
    res["config2_q_assembly_frac_of_hbm_peak"] = res["config2_q_assembly_GBps"] / HBM_PEAK_GBPS
    del Q
    ix = system.indices_device
    # the reference's A (every entry, solve_film.py:296-305) and the form the Cholesky consumes (diag(w) A, lower tiles)
    t = timed(lambda: kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix, ix, sign=-1.0,
                                              dtype="float64"))
    res["config2_a_assembly_ms"] = t
    res["config2_a_assembly_GBps"] = ni * ni * 8 / (t * 1e-3) / 1e9
    model = None
    torch.cuda.empty_cache()
    model = sc.factorize_model(device=device, current_units="uA")            # untimed: the allocator takes its blocks
    tf = []
    for _ in range(3):
        model = None
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model = sc.factorize_model(device=device, current_units="uA")
        torch.cuda.synchronize()
        tf.append((time.perf_counter() - t1) * 1e3)
    res["config2_factorize_ms"] = float(np.median(tf))
    flops = ni ** 3 / 3.0 if used_chol else 2.0 * ni ** 3 / 3.0
    res["config2_factorization_TFLOPs"] = flops / (res["config2_factorize_ms"] * 1e-3) / 1e12
    res["config2_factorization_route"] = "cholesky of diag(w) A" if used_chol else "lu of -A"
    sc.solve(model=model, applied_field=sc.ConstantField(1.0), progress_bar=False)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sc.solve(model=model, applied_field=sc.ConstantField(1.0), progress_bar=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t1) * 1e3)
    res["config2_solve_film_ms"] = float(np.median(ts))
    res["config2_cold_solves_per_s"] = 1e3 / (res["config2_factorize_ms"] + res["config2_solve_film_ms"])
    del model
    torch.cuda.empty_cache()
    return res


def config3_two_films(sc, torch, iterations):
    """Config 3: washer + shield disk, 19 927 vertices per film (K = 81), one GPU: cold self-consistent solves with
    the fixed iteration count and "to convergence" (max|dg|/max|g| < 1e-8; the reference has no such test)."""
    from superscreen_amd import synthetic

    K3 = int(os.environ.get("BENCH_CONFIG3_K", "81"))             # testing aid: smaller meshes
    device = synthetic.make_stack_device(K3, ("washer", "disk"), solve_dtype="float64")

    def cold(**kw):
        model = sc.factorize_model(device=device, current_units="uA", expected_passes=kw.get("iterations", 0) + 1)
        return sc.solve(model=model, applied_field=sc.ConstantField(1.0), progress_bar=False, **kw)

    cold(iterations=iterations)                                    # warm-up
    res = {"config3_vertices_per_film": len(device.meshes[list(device.films)[0]].sites)}
    for label, kw in (("10iter", dict(iterations=iterations)), ("to_1e-8", dict(iterations=200, tolerance=1e-8))):
        ts, sols = [], None
        for _ in range(3):
            sols = None
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sols = cold(**kw)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        res[f"config3_solves_per_s_{label}"] = 1.0 / float(np.median(ts))
        if label == "to_1e-8":
            res["config3_iterations_to_1e-8"] = len(sols) - 1
    torch.cuda.empty_cache()
    return res


def configH_float32(sc, torch, K, iterations, steps=6):
    """Config H in the reference's DEFAULT precision (``Device(..., solve_dtype="float32")``, device/device.py:57): the
    same cold self-consistent solve as the headline, float32 factor and solves (sheet currents and coupling sums stay
    float64 as in the reference, solve.py:508-515).  The factorization is priced against the FP32 matrix peak."""
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(K, ("washer", "disk"), solve_dtype="float32")

    def cold(field):
        model = sc.factorize_model(device=device, current_units="uA", expected_passes=iterations + 1)
        return model, sc.solve(model=model, applied_field=sc.ConstantField(field), field_units="mT",
                               iterations=iterations, progress_bar=False)

    model = sols = None
    for i in range(2):
        model = sols = None
        model, sols = cold(0.1 * (i + 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        model = sols = None
        model, sols = cold(0.1 * (i + 3))
    torch.cuda.synchronize()
    ms_step = (time.perf_counter() - t0) / steps * 1e3
    assert len(sols) == iterations + 1 and sols[-1].film_solutions["disk1"].stream.dtype == np.float32
    unknowns = [int(len(s_.indices)) for s_ in model.film_systems.values()]
    used_chol = all(s_.chol is not None for s_ in model.film_systems.values())
    tf = []
    for _ in range(5):
        model = sols = None
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model = sc.factorize_model(device=device, current_units="uA")
        torch.cuda.synchronize()
        tf.append((time.perf_counter() - t1) * 1e3)
    t_fact = float(np.median(tf))
    flops = sum(u ** 3 / 3.0 for u in unknowns) * (1.0 if used_chol else 2.0)
    del model, sols
    torch.cuda.empty_cache()
    return {"configH_float32_ms_per_step": ms_step, "configH_float32_solves_per_s": 1e3 / ms_step,
            "configH_float32_factorization_ms": t_fact,
            "configH_float32_factorization_TFLOPs": flops / (t_fact * 1e-3) / 1e12,
            "configH_float32_factorization_frac_of_fp32_matrix_peak": flops / (t_fact * 1e-3) / 1e12 / 157.3,
            "configH_float32_note": "reference default solve_dtype; parity of this precision: "
                                    "tests/test_headline_gpu.py::test_configH_float32_no_worse_than_the_reference_in_float32"}


def configH_mixed(sc, torch, device, iterations, ref_sols, ref_field_mT, steps=6):
    """Config H with ``factorize_model(method="mixed")`` -- an opt-in extension, NOT the headline (which stays pure
    float64): diag(w) A factored in FLOAT32, every solve refined to float64 against the matrix-free system
    (superscreen_amd/solver.py::_mixed_solve).  Reports the cold step and, for the applied field of the headline's last
    timed step, its distance from that float64 solution (stream functions of every iterate, the fluxoid ring of
    ``fluxoid_parity``) -- the headline's own distance from the CPU oracle is ``parity`` (1e-13)."""
    from superscreen_amd import Polygon, synthetic

    def cold(field):
        model = sc.factorize_model(device=device, current_units="uA", method="mixed", expected_passes=iterations + 1)
        return model, sc.solve(model=model, applied_field=sc.ConstantField(field), field_units="mT", iterations=iterations,
                               progress_bar=False)

    model = sols = None
    for i in range(2):
        model = sols = None
        model, sols = cold(0.1 * (i + 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        model = sols = None
        model, sols = cold(0.1 * (i + 3))
    torch.cuda.synchronize()
    ms_step = (time.perf_counter() - t0) / steps * 1e3
    tf = []
    for _ in range(5):
        model = sols = None
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model = sc.factorize_model(device=device, current_units="uA", method="mixed")
        torch.cuda.synchronize()
        tf.append((time.perf_counter() - t1) * 1e3)
    t1 = time.perf_counter()
    sols = sc.solve(model=model, applied_field=sc.ConstantField(ref_field_mT), field_units="mT", iterations=iterations,
                    progress_bar=False)
    torch.cuda.synchronize()
    t_solve = (time.perf_counter() - t1) * 1e3
    err_g = err_f = 0.0
    names = list(ref_sols[0].film_solutions)
    film_poly = device.films[names[0]].points
    r_film = float(np.max(np.hypot(film_poly[:, 0], film_poly[:, 1])))
    holes = list(device.holes.values())
    r_hole = float(np.max(np.hypot(holes[0].points[:, 0], holes[0].points[:, 1]))) if holes else 0.0
    ring = Polygon(points=synthetic.circle_points(0.5 * (r_hole + r_film), 301)).points
    for a, b in zip(sols, ref_sols):
        for nm in names:
            ga, gb = a.film_solutions[nm].stream, b.film_solutions[nm].stream
            err_g = max(err_g, float(np.max(np.abs(ga - gb)) / np.max(np.abs(gb))))
            fa = a.polygon_fluxoid(ring, film=nm, units="mT * um**2", with_units=False)
            fb = b.polygon_fluxoid(ring, film=nm, units="mT * um**2", with_units=False)
            scale = max(abs(fb.flux_part), abs(fb.supercurrent_part))
            err_f = max(err_f, max(abs(fa.flux_part - fb.flux_part), abs(fa.supercurrent_part - fb.supercurrent_part)) / scale)
    del model, sols
    torch.cuda.empty_cache()
    return {"configH_mixed_ms_per_step": ms_step, "configH_mixed_solves_per_s": 1e3 / ms_step,
            "configH_mixed_factorization_ms": float(np.median(tf)), "configH_mixed_solve_ms": t_solve,
            "configH_mixed_max_rel_err_stream_vs_float64": err_g, "configH_mixed_max_rel_err_fluxoid_vs_float64": err_f,
            "configH_mixed_note": "opt-in factorize_model(method='mixed'): float32 Cholesky + 2 float64 refinement sweeps per "
                                  "solve against the matrix-free system; errors = distance from the headline's float64 "
                                  "solution of the same field, all iterates (north_star tolerance 1e-6); parity vs the "
                                  "oracle: tests/test_solve_gpu.py::test_mixed_precision_factorization_refined_to_float64"}


def pipelined_cold_solves(sc, torch, device, iterations, steps=8):
    """Throughput of back-to-back cold solves when solve i and the factorization of solve i + 1 overlap (what a scan
    over devices or geometries can do; the headline stays the sequential step): the passes of a solve are bound by
    HBM (GEMV chains, matrix pipes idle), a factorization by the matrix pipes.  Two models alive, two host threads,
    each phase on a stream of its own; a model changes hands after a host-side wait for its factorization."""
    import threading

    s_fact, s_solve = torch.cuda.Stream(), torch.cuda.Stream()
    errors = []

    def factorize(box):
        try:
            with torch.cuda.stream(s_fact):
                box.append(sc.factorize_model(device=device, current_units="uA"))
            s_fact.synchronize()
        except BaseException as exc:   # noqa: BLE001 -- reported by the caller
            errors.append(exc)

    def solve(model, field, box):
        try:
            with torch.cuda.stream(s_solve):
                box.append(sc.solve(model=model, applied_field=sc.ConstantField(field), field_units="mT",
                                    iterations=iterations, progress_bar=False))
            s_solve.synchronize()
        except BaseException as exc:   # noqa: BLE001
            errors.append(exc)

    def run(n):
        box = []
        factorize(box)
        model, done = box[0], 0
        for i in range(n):
            nxt, out = [], []
            threads = [threading.Thread(target=solve, args=(model, 0.1 * (i + 1), out))]
            if i + 1 < n:
                threads.append(threading.Thread(target=factorize, args=(nxt,)))
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            assert len(out[0]) == iterations + 1
            done += 1
            model = nxt[0] if nxt else None
        return done

    run(2)                                                         # warm-up: streams, chain-stream calibration, allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = run(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    torch.cuda.empty_cache()
    return {"pipelined_cold_solves_per_s": n / elapsed, "pipelined_cold_solve_ms": elapsed / n * 1e3,
            "pipelined_note": f"{n} cold solves, factorization of solve i + 1 beside the passes of solve i (two streams, two "
                              "models alive); includes the first factorization, which overlaps nothing"}


def configH_alt_2x50k(sc, torch, iterations):
    """The other reading of "50k-vertex 2-film device": 50 311 vertices PER film (K = 129 washer + disk,
    100 622 vertices in all): one cold self-consistent solve."""
    from superscreen_amd import synthetic

    Ka = int(os.environ.get("BENCH_CONFIGH_ALT_K", "129"))        # testing aid: smaller meshes
    device = synthetic.make_stack_device(Ka, ("washer", "disk"), solve_dtype="float64")

    def cold():
        model = sc.factorize_model(device=device, current_units="uA", expected_passes=iterations + 1)
        unknowns = [int(len(s_.indices)) for s_ in model.film_systems.values()]
        return unknowns, sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=iterations,
                                  progress_bar=False)

    cold()                                                         # warm-up (allocator: 2 x 14 GB factors)
    ts = []
    for _ in range(2):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        unknowns, sols = cold()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t1)
        assert len(sols) == iterations + 1
        del sols
    t = float(np.median(ts))
    torch.cuda.empty_cache()
    return {"configH_alt_2x50k_vertices_per_film": len(device.meshes[list(device.films)[0]].sites),
            "configH_alt_2x50k_unknowns": unknowns, "configH_alt_2x50k_ms_per_solve": t * 1e3,
            "configH_alt_2x50k_solves_per_s": 1.0 / t,
            "configH_alt_2x50k_TFLOPs_factorization_flops_only": sum(u ** 3 / 3.0 for u in unknowns) / t / 1e12}


def configH_own_meshes(sc, torch, iterations):
    """Config H's shape on films that share NOTHING: a washer on a K = 91 mesh (25 117 vertices) below a smaller
    disk on a K = 85 mesh (21 931 vertices, radius 4.5 um) that sits off the axis, different Lambda per layer -- the
    reference's general case (every film its own mesh, ``solver/solve.py:495-515``); parity of this device class:
    ``tests/test_solve_gpu.py::test_films_with_their_own_meshes_*`` (reference fixture + oracle)."""
    from superscreen_amd import synthetic

    Ka = int(os.environ.get("BENCH_OWN_MESHES_K", "91"))          # testing aid: smaller meshes
    spec = dict(layers=[dict(name="layer0", z0=0.0, Lambda=0.1), dict(name="layer1", z0=0.5, Lambda=0.05)],
                films=[dict(name="washer", kind="washer", K=Ka, layer="layer0", film_radius=5.0),
                       dict(name="shield", kind="disk", K=max(3, Ka - 6), layer="layer1", film_radius=4.5,
                            center=(0.4, -0.3))])
    device = synthetic.make_device(spec["films"], spec["layers"])
    field = sc.Parameter(synthetic.tilted_field, B0=1.0)

    def cold():
        model = sc.factorize_model(device=device, current_units="uA", expected_passes=iterations + 1)
        return [int(len(s_.indices)) for s_ in model.film_systems.values()], \
            sc.solve(model=model, applied_field=field, field_units="mT", iterations=iterations, progress_bar=False)

    cold()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        unknowns, sols = cold()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t1)
        assert len(sols) == iterations + 1
        del sols
    t = float(np.median(ts))
    torch.cuda.empty_cache()
    return {"configH_own_meshes_vertices": [len(device.meshes[f].sites) for f in device.films],
            "configH_own_meshes_unknowns": unknowns, "configH_own_meshes_ms_per_solve": t * 1e3,
            "configH_own_meshes_solves_per_s": 1.0 / t,
            "configH_own_meshes_TFLOPs_factorization_flops_only": sum(u ** 3 / 3.0 for u in unknowns) / t / 1e12}


# ---------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5 on N ranks
# ---------------------------------------------------------------------------------------------------
def config4_scan(sc, torch, dist, rank, world, iterations):
    """64-value applied-field scan of the config-3 device (washer + shield disk, K = 81: 2 x 19 927 vertices)
    through parallel.solve_sweep_sharded.  Every rank factorizes its own replica inside the timed region."""
    from superscreen_amd import synthetic
    from superscreen_amd.parallel import FilmPlacement, SweepGrid, solve_sweep_grid, solve_sweep_sharded

    K4 = int(os.environ.get("BENCH_CONFIG4_K", "81"))             # testing aid: smaller meshes
    device = synthetic.make_stack_device(K4, ("washer", "disk"), solve_dtype="float64")
    scan = [float(v) for v in np.linspace(0.1, 6.4, 64)]          # mT

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def run(fields, shard):
        model = sc.factorize_model(device=device, current_units="uA")
        if shard:
            out = solve_sweep_sharded(model, fields, rank=rank, world=world, iterations=iterations, all_iterations=False)
            local = out[2]
        else:
            local = sc.solve_sweep(model, fields, iterations=iterations, all_iterations=False)
        torch.cuda.synchronize()
        return len(local)

    run(scan[:max(2, 64 // world)], False)                        # warm-up: allocator, lazy initialisation
    res = {}
    for label, shard in (("strong", True), ("weak", False)):
        barrier()
        t0 = time.perf_counter()
        n_local = run(scan, shard)
        barrier()
        elapsed = time.perf_counter() - t0
        total = n_local
        if dist is not None:
            tt = torch.tensor([elapsed, float(n_local)], dtype=torch.float64, device="cuda")
            tmax = tt.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            elapsed, total = float(tmax[0].item()), int(round(tt[1].item()))
        res[f"config4_{label}_fields_total"] = total
        res[f"config4_{label}_seconds"] = elapsed
        res[f"config4_{label}_solves_per_s"] = total / elapsed
    res["config4_note"] = ("config-3 device, 64 fields linspace(0.1, 6.4) mT, 10 iterations, final iterate returned as "
                           "Solutions; strong = 64 fields over the ranks, weak = 64 fields per rank; each rank's "
                           "factorization is inside the timed region; no data-path collective")
    # (film owner) x (field shard) grid: a rank factors ONE film and sweeps 64 / shards columns of it; one all-reduce of
    # the [n, nvec] result arrays per pass inside a shard's group (parallel.SweepGrid).  Cold (factorization inside
    # the timed region) and with the pre-factorized model (the reference pattern: one factorize_model, many solves).
    if world > 1:
        grid = SweepGrid(len(device.films), rank=rank, world=world)
        solve_sweep_grid(device, scan[:max(2, 64 // world)], grid, iterations=iterations, all_iterations=False)
        for label in ("cold", "prefactorized"):
            model = None
            if label == "prefactorized":
                model = solve_sweep_grid(device, scan[:2], grid, iterations=iterations, all_iterations=False)[3]
            barrier()
            t0 = time.perf_counter()
            b, e, local, _ = solve_sweep_grid(device, scan, grid, model=model, iterations=iterations, all_iterations=False)
            torch.cuda.synchronize()
            barrier()
            elapsed = time.perf_counter() - t0
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            res[f"config4_grid_{label}_seconds"] = float(tmax.item())
            res[f"config4_grid_{label}_solves_per_s"] = 64 / float(tmax.item())
            del local, model
        res["config4_grid_layout"] = f"{grid.film_ranks} film owners x {grid.shards} field shards"
    else:
        # what a rank of the grid does on an 8-GPU node, measured on this GPU: one film's factorization, and the
        # scan of the pre-factorized model for 64 fields (one GPU) and for 16 fields (one shard of four)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            m1 = sc.factorize_model(device=device, current_units="uA", placement=FilmPlacement(rank=1, world=2))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
            del m1
        res["config4_one_film_factorize_ms"] = float(np.median(ts)) * 1e3
        model = sc.factorize_model(device=device, current_units="uA")
        for nf in (64, 16):
            sc.solve_sweep(model, scan[:nf], iterations=iterations, all_iterations=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sc.solve_sweep(model, scan[:nf], iterations=iterations, all_iterations=False)
            torch.cuda.synchronize()
            res[f"config4_prefactorized_{nf}_fields_ms"] = (time.perf_counter() - t1) * 1e3
        del model
    return res


def config5_stack(sc, torch, dist, rank, world, iterations, steps=2):
    """Cold self-consistent solve of the 4-film stack (4 disks on the K = 100 mesh: 4 x 30 301 vertices, z = 0,
    0.5, 1.0, 1.5 um) on `world` ranks."""
    from superscreen_amd import synthetic
    from superscreen_amd.parallel import FilmPlacement

    K5 = int(os.environ.get("BENCH_CONFIG5_K", "100"))            # testing aid: smaller meshes
    device = synthetic.make_stack_device(K5, ("disk",) * 4, z_spacing=0.5, solve_dtype="float64")
    films = list(device.films)
    placement = coupling = None
    if world > 1 and world < 2 * len(films):
        placement, mode = FilmPlacement(rank=rank, world=world), "FilmPlacement (owner computes; one all-reduce of the result vectors per pass)"
        if world > len(films):
            mode += f"; ranks {len(films)}..{world - 1} own no film and idle (fewer than two ranks per film: no helper groups)"
    elif world > 1:
        # more ranks than films: one group of world // n_films ranks per film -- the owner factors and solves, every
        # rank of the group takes a source slice of the coupling sums whose target is the group's film (summed inside
        # the group); replicating the four factorizations on eight ranks (CouplingPlan) would be ~ 3 x slower than
        # four owners
        placement = FilmPlacement(rank=rank, world=world, n_films=len(films))
        mode = (f"FilmPlacement with helper groups ({len(films)} groups of {placement.group_size} ranks: owner factors / solves, "
                "coupling sums of the group's film split by source slice and summed inside the group; one all-reduce of the "
                "result vectors per pass across groups)")
    else:
        mode = "single GPU"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def step():
        model = sc.factorize_model(device=device, current_units="uA", placement=placement, expected_passes=iterations + 1)
        sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=iterations,
                        placement=placement, coupling=coupling)
        return model, sols

    model, sols = step()                                           # warm-up
    model = sols = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        model = sols = None
        model, sols = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    g = sols[-1].film_solutions[films[0]].stream
    assert np.isfinite(g).all() and len(sols) == iterations + 1
    checksum = float(np.abs(g).max())
    out = {"config5_solves_per_s": steps / elapsed, "config5_ms_per_solve": elapsed / steps * 1e3,
           "config5_mode": mode, "config5_vertices_per_film": len(device.meshes[films[0]].sites),
           "config5_max_abs_stream_film0": checksum}
    if world == 1:
        # what ONE rank of an N-GPU run does, measured here on one GPU (DESIGN.md section 6 derives the projected
        # N = 2 / 4 / 8 times from these): one film's factorization, one film's pass, the coupling sums of one
        # target film from all three sources and from half of every source (an owner / helper pair at N = 8)
        from superscreen_amd import kernels

        def timed(fn, reps=5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            ts = []
            for _ in range(reps):
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return float(np.median(ts))

        fds, info = model.film_data, model.film_info
        tgt = films[0]
        t, J = fds[tgt], {f: torch.from_numpy(np.ascontiguousarray(sols[-1].film_solutions[f].current_density)).cuda()
                          for f in films}
        buf = torch.zeros(t.n, dtype=t.tdtype, device="cuda")

        def couple(frac):
            for src in films[1:]:
                sd = fds[src]
                lo, hi = sd.src_range
                kernels.biot_savart(sd.xy, sd.w_t, J[src], t.xy, info[tgt].z0 - info[src].z0, buf, accumulate=True,
                                    src_begin=lo, src_end=lo + int((hi - lo) * frac))

        out["config5_one_target_coupling_ms"] = timed(lambda: couple(1.0))
        out["config5_one_target_half_sources_coupling_ms"] = timed(lambda: couple(0.5))
        del model, sols, J, buf
        torch.cuda.empty_cache()
        one = synthetic.make_stack_device(K5, ("disk",), solve_dtype="float64")
        tf = []
        m1 = sc.factorize_model(device=one, current_units="uA")
        for _ in range(3):
            m1 = None
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            m1 = sc.factorize_model(device=one, current_units="uA")
            torch.cuda.synchronize()
            tf.append((time.perf_counter() - t1) * 1e3)
        out["config5_one_film_factorize_ms"] = float(np.median(tf))
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sc.solve(model=m1, applied_field=sc.ConstantField(1.0), iterations=0, progress_bar=False)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t1) * 1e3)
        out["config5_one_film_pass_ms"] = float(np.median(ts[1:]))
        del m1
    else:
        del model, sols
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path.")
    # BENCH_SHARE_GPU=1 (testing aid for single-GPU boxes): all ranks use cuda:0 and gloo carries the
    # collectives, so that the N > 1 control flow can be exercised without N GPUs
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
        warm = torch.ones(1, device="cuda")
        dist.all_reduce(warm)                                      # communicator set-up outside every timed region
        torch.cuda.synchronize()

    import superscreen_amd as sc
    from superscreen_amd import _hip, kernels, synthetic

    lib = _hip.load_library()
    device = synthetic.make_stack_device(args.K, ("washer", "disk"), solve_dtype="float64")
    n = len(device.meshes["washer0"].sites)

    def step(i):
        field = 0.1 * (1 + rank + world * i)  # mT; every rank / step solves a different field
        # a cold solve: the factorization serves this solve's iterations + 1 passes and says so, which is what the
        # reference's plain `solve(device=...)` call does here too (solver.py: expected_passes -> 2048-row solve blocks)
        model = sc.factorize_model(device=device, current_units="uA", expected_passes=args.iterations + 1)
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
    # Live HIP-event bracketing of the two candidates for the dominant kernel (the trailing updates: kinds 0 and 1,
    # ~ 100 launches per step).  The ~ 1300 short products of the panel chains (kind 2) are bracketed in a separate
    # factorization below: their event pairs sit on the latency-critical chain streams and cost the step 7 %.
    _hip.check(lib.ssa_profile_begin_kinds(0b011), "ssa_profile_begin_kinds")
    t0 = time.perf_counter()
    for i in range(args.steps):
        model = sols = None
        model, sols = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    parity_sols, parity_field = sols, 0.1 * (1 + rank + world * (args.steps - 1))   # last timed step, for the oracle
    parity_model_solve_block = max((s.chol.solve_block for s in model.film_systems.values() if s.chol is not None), default=0)
    prof = {}
    for kind, label in ((0, "ssa::gemm_kernel<double, true> (NN: LU trailing / in-panel updates)"),
                        (1, "ssa::gemm_op_kernel<double, 0, 1, true> (SYRK on the lower tiles: Cholesky trailing update)"),
                        (2, "ssa::gemm_op_kernel<double, 0, 1, false> (Cholesky strips and L21 = A21 W^T panel products)")):
        ms, fl, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        _hip.check(lib.ssa_profile_read(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt)),
                   "ssa_profile_read")
        prof[label] = (ms.value, fl.value, cnt.value)
    _hip.check(lib.ssa_profile_end(), "ssa_profile_end")
    # the chains' strips and panel products: one more factorization (untimed) with only that kind bracketed
    model = None
    _hip.check(lib.ssa_profile_begin_kinds(0b11100), "ssa_profile_begin_kinds")
    model = sc.factorize_model(device=device, current_units="uA")
    torch.cuda.synchronize()
    chain_label = [k for k in prof if "strips" in k][0]
    for kind, label in ((2, chain_label),
                        (3, "ssa::chol_tail_round_kernel<double> (rounds of the last 10 240 columns: the films' diagonal-block "
                            "kernels + the lower tiles of their pending updates in one launch; flops of the tiles)"),
                        (4, "ssa::gemm_nt_small_batch_kernel<double> (the rounds' batched L21 = A21 W^T and next-block-column "
                            "products)")):
        ms, fl, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        _hip.check(lib.ssa_profile_read(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt)), "ssa_profile_read")
        prof[label] = (ms.value, fl.value, cnt.value)
    _hip.check(lib.ssa_profile_end(), "ssa_profile_end")
    labels = list(prof)
    dom_label = max(labels[:2], key=lambda k: prof[k][0])
    gemm_ms, gemm_fl, gemm_n = prof[dom_label]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    total_solves = args.steps * world
    value = total_solves / elapsed
    unknowns = [int(len(s.indices)) for s in model.film_systems.values()]
    used_chol = all(s.chol is not None for s in model.film_systems.values())
    solve_block = parity_model_solve_block

    extras = {}
    if rank == 0:
        # sanity: the last step's answer is a converged, finite screening solution
        g = sols[-1].film_solutions["disk1"].stream
        assert np.isfinite(g).all() and len(sols) == args.iterations + 1
    if rank == 0 and not args.no_extras:
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
            for _ in range(4):     # back to back: the device-side rate (no idle behind the event while the host launches)
                kernels.q_assemble(fd.xy, fd.w, C, "float64", out=Q, ld=ld)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3 / 4)
        tq = float(np.median(ts))
        extras["q_assembly_GBps"] = n * n * 8 / tq / 1e9
        extras["q_assembly_ms"] = tq * 1e3
        extras["q_assembly_frac_of_hbm_peak"] = extras["q_assembly_GBps"] / HBM_PEAK_GBPS
        extras["q_assembly_timing"] = "4 back-to-back ssa_q_assemble calls between two events, / 4; median of 5"
        # the plain fill kernel on the same buffer: what a pure store stream reaches on this box
        kernels.fill_probe(Q)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(4):
                kernels.fill_probe(Q)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3 / 4)
        extras["fill_probe_GBps"] = Q.numel() * 8 / float(np.median(ts)) / 1e9
        del Q
        # the factorization alone (assembly + Cholesky of both films; the step's first phase)
        tf = []
        for _ in range(3):
            model = sols = None
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model = sc.factorize_model(device=device, current_units="uA")
            torch.cuda.synchronize()
            tf.append(time.perf_counter() - t1)
        t_fact = float(np.median(tf))
        fact_flops = sum(u ** 3 / 3.0 for u in unknowns) if used_chol else sum(2.0 * u ** 3 / 3.0 for u in unknowns)
        extras["factorization_ms"] = t_fact * 1e3
        # what a dependent launch costs on the panel-chain streams the schedule uses, as the library measured them on
        # this device against the caller's stream (chain_streams.hip); the first entries are the
        # ones in use
        chain_us, chain_pipe = kernels.chol_chain_stream_costs()
        extras["chain_stream_dependent_launch_us"] = [round(v, 1) for v in chain_us]
        extras["chain_stream_pipe_group"] = chain_pipe
        extras["factorization_TFLOPs"] = fact_flops / t_fact / 1e12
        extras["factorization_frac"] = fact_flops / t_fact / 1e12 / FP64_MFMA_PEAK_TFLOPS
        extras["step_TFLOPs_factorization_flops_only"] = fact_flops / (elapsed / args.steps) / 1e12
        # what a bare register-only v_mfma_f64 stream (8 waves per SIMD, no memory, no LDS) is held at on
        # this box right now: 45-62 TFLOP/s observed -- the chip lowers its clock under a pure MFMA load,
        # so this is a power-management reading, not a ceiling (the SYRK engine itself reaches 67 TFLOP/s
        # at K = 2048, tools/probes/syrk_k_probe.py); the roofline fraction is priced against 78.6
        extras["fp64_mfma_register_only_TFLOPs"] = kernels.mfma_probe(4000)
        # field map above the device: 512 x 512 image, all-pairs Biot-Savart of one film
        sols = sc.solve(model=model, applied_field=sc.ConstantField(0.3), iterations=args.iterations)
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
        # a 64-value applied-field scan of the SAME (config H) device carried as the columns of one
        # multi-right-hand-side solve (solve_sweep; final iterate of every field returned as Solutions)
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
        extras["configH_iterations_to_1e-8"] = len(conv) - 1
        del conv
    model = sols = None
    torch.cuda.empty_cache()
    def build_line():
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        avg_us = gemm_ms * 1e3 / max(1, gemm_n)
        avg_gflop = gemm_fl / max(1, gemm_n) / 1e9
        is_syrk = "gemm_op_kernel" in dom_label
        traffic_doc, traffic_src = load_profile(PMC_TRAFFIC_FILES) if is_syrk else (None, None)
        mfma_doc, mfma_src = load_profile(PMC_MFMA_FILES) if is_syrk else (None, None)
        traffic = float(traffic_doc["traffic_bytes_per_launch"]) if traffic_doc else None
        stale = None
        if traffic_doc is not None:
            # the PMC summary records what the profiled run looked like; a kernel that has changed since
            # shows up as a different average launch (duration or flops)
            ref_us, ref_gf = traffic_doc.get("avg_launch_us"), traffic_doc.get("avg_launch_gflop")
            if ref_us is None or ref_gf is None:
                stale = True
            else:
                # flops per launch: 5 % (the schedule); duration: 15 % (the boxes of this pool differ by 12 % on
                # the same binary, MI355X_MICROARCH.md "Devices differ")
                stale = bool(abs(avg_gflop / ref_gf - 1) > 0.05 or abs(avg_us / ref_us - 1) > 0.15)
        strip_ms, strip_fl, strip_n = prof[labels[2]]
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
                "unknowns": unknowns,
                "iterations": args.iterations,
                "parallelism": f"independent solves x{world} (one per rank and step; no data-path collective)",
                # the reference evaluates self_field = Q @ (w g) (solve_film.py:565); the passes of this line take it
                # from the solved London system on the unknowns' rows and from the all-pairs sum on the others
                # (factorize_model(self_field="auto"): float64 and uniform Lambda only; other devices and
                # self_field="matrix_free" pay the all-pairs sum on every row).  It is an output, not an input of the
                # next iterate; its difference from the reference's value is parity.max_rel_err_self_field.
                "self_field_mode": "auto: London identity on the unknowns' rows, all-pairs sum on the other rows",
                # every step factorizes for ITS passes: factorize_model(expected_passes=iterations + 1), what the
                # reference's plain cold call solve(device=...) does in this package -- the triangular solves then run
                # on 2048-row diagonal blocks (a model made for reuse: 4096)
                "factorization": ("cholesky of diag(w) A" if used_chol else "lu of -A") +
                                 f", solve blocks of {solve_block} rows",
            },
            "roofline": {
                "kernel": dom_label + ", v_mfma_f64_16x16x4_f64",
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": (f"{traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; "
                                   "bytes per launch, fetch x2 per MI355X_MICROARCH.md)") if traffic_src else None,
                "traffic_stale": stale,
                "algorithmic_bytes_per_launch": chol_schedule(unknowns)[0],
                "launches": int(gemm_n),
                "avg_launch_us": avg_us,
                "avg_launch_gflop": avg_gflop,
                "mfma_pipe": ({"mfma_busy_fraction_of_active_cycles": float(mfma_doc["mfma_busy_fraction_of_active_cycles"]),
                               "effective_clock_GHz": float(mfma_doc["effective_clock_GHz"]),
                               "source": f"{mfma_src} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                                         "SQ_INSTS_VALU_MFMA_MOPS_F64 pass of this command)"} if mfma_doc else None),
                "factorization_frac": extras.get("factorization_frac"),
                "other_kernels": {
                    lab: {"launches": int(prof[lab][2]), "avg_launch_us": prof[lab][0] * 1e3 / max(1, prof[lab][2]),
                          "TFLOPs": (prof[lab][1] / (prof[lab][0] * 1e-3) / 1e12) if prof[lab][0] > 0 else 0.0,
                          "measured": "one factorization outside the timed region"}
                    for lab in labels[2:]
                },
            },
            "extras": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(args.K, args.iterations, args.cpu_budget_s, parity_sols, parity_field)
            out["parity"] = base.pop("parity")
            out["cpu_baseline"] = base
            out["extras"]["gpu_over_cpu_baseline"] = value / base["value"]
        return out

    def guarded(name, fn):
        """An extra that fails costs its own numbers, never the line."""
        try:
            return fn()
        except Exception as exc:   # noqa: BLE001 -- reported in the line
            torch.cuda.empty_cache()
            return {f"{name}_error": f"{type(exc).__name__}: {exc}"[:300]}

    if rank == 0 and not args.no_extras:
        # BASELINE configs 2 and 3 and the 2 x 50 311 reading of the headline, on this rank's GPU
        extras.update(guarded("config3", lambda: config3_two_films(sc, torch, args.iterations)))
        extras.update(guarded("config2", lambda: config2_single_film(sc, torch, kernels)))
        extras.update(guarded("pipelined", lambda: pipelined_cold_solves(sc, torch, device, args.iterations)))
        extras.update(guarded("configH_float32", lambda: configH_float32(sc, torch, args.K, args.iterations)))
        extras.update(guarded("configH_mixed", lambda: configH_mixed(sc, torch, device, args.iterations, parity_sols,
                                                                     parity_field)))
        extras.update(guarded("configH_alt_2x50k", lambda: configH_alt_2x50k(sc, torch, args.iterations)))
        extras.update(guarded("configH_own_meshes", lambda: configH_own_meshes(sc, torch, args.iterations)))

    def emit_line():
        if rank == 0:
            print(json.dumps(build_line()), flush=True)

    if not args.no_extras:
        # BASELINE configs 4 and 5 on all ranks (collective calls: every rank takes part).  With more than one rank a
        # failure on ONE rank leaves the others inside a collective: a watchdog then prints the line as it stands
        # (headline measured, collective extras marked) and ends the process, so that the record of the run survives.
        watchdog = None
        if world > 1:
            import threading

            limit = float(os.environ.get("BENCH_COLLECTIVE_EXTRAS_TIMEOUT_S", "420"))

            def give_up(reason=None):
                # The headline was measured before the collective extras started, so the line still goes out -- but
                # a process that stalled or was killed must NOT look like a success: exit status 3.
                extras["collective_extras_error"] = reason or (
                    f"configs 4 / 5 did not finish within {limit:.0f} s on {world} ranks")
                emit_line()
                os._exit(3)

            watchdog = threading.Timer(limit, give_up)
            watchdog.daemon = True
            watchdog.start()
            if rank == 0:
                import signal

                # the launcher ends the surviving ranks with SIGTERM when one of them dies: the line goes out first
                signal.signal(signal.SIGTERM, lambda *_: give_up(
                    f"terminated by the launcher (SIGTERM) during configs 4 / 5 on {world} ranks"))
        for name, fn in (("config4", config4_scan), ("config5", config5_stack)):
            res = guarded(name, lambda: fn(sc, torch, dist, rank, world, args.iterations))
            if rank == 0:
                extras.update(res)
            torch.cuda.empty_cache()
        if watchdog is not None:
            watchdog.cancel()
            if rank == 0:
                signal.signal(signal.SIGTERM, signal.SIG_DFL)

    emit_line()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
