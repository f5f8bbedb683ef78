"""Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box and writes the summaries that
get committed under profiles/ (run from the repo root on the box; development aid).

    python tools/collect_profiles.py <tag> [summary_dir]

Passes (each `rocprofv3 ... -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras`, counters in their
own passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; the HBM-bound kernels
from two more passes of tools/hbm_kernels.py):
  stats   --kernel-trace --stats                      -> <tag>_bench_kernel_stats.csv, <tag>_bench.json
  fetch   --pmc FETCH_SIZE --kernel-trace
  write   --pmc WRITE_SIZE --kernel-trace             -> <tag>_syrk_pmc.json, <tag>_hbm_pmc.json
  mfma    --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace
                                                      -> <tag>_syrk_mfma_pmc.json
and one plain `python3 bench.py` (no profiler)        -> <tag>_bench_full.json
Corrections: counters in KiB; FETCH_SIZE x2 on gfx950 (half-count of wide reads); WRITE_SIZE exact.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYRK = "gemm_op_kernel<double, 0, 1, true>"
SIMDS, XCDS = 256 * 4, 8


def run_pass(name, flags, out_dir, bench_args, script="bench.py"):
    d = os.path.join(out_dir, name)
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3"] + flags + ["--output-format", "csv", "-d", d, "-o", name, "--", "python3",
           os.path.join(ROOT, script)] + bench_args
    p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not line:
        sys.stderr.write(p.stdout[-2000:] + p.stderr[-4000:])
        raise SystemExit(f"pass {name} failed")
    return d, json.loads(line[-1])


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[0]


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def counter_per_kernel(path, counter, needle):
    vals = [float(r["Counter_Value"]) for r in rows(path) if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return vals


def main():
    tag = sys.argv[1]
    prof = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "profiles_" + tag))
    os.makedirs(prof, exist_ok=True)
    out_dir = os.path.join("/tmp", "ssa_profiles_" + tag)   # raw rocprofv3 output (hundreds of MB) stays on the box
    os.makedirs(out_dir, exist_ok=True)
    # the headline line only (--no-extras): every launch of the profiled run belongs to the timed workload
    args = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]

    d, line = run_pass("stats", ["--kernel-trace", "--stats"], out_dir, args)
    shutil.copy(find(d, "kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
    json.dump(line, open(os.path.join(prof, f"{tag}_bench.json"), "w"), indent=1)
    stat = [r for r in rows(find(d, "kernel_stats.csv")) if SYRK in r["Name"]][0]
    avg_us_stats = float(stat["AverageNs"]) * 1e-3
    roof = line["roofline"]

    df, _ = run_pass("fetch", ["--pmc", "FETCH_SIZE", "--kernel-trace"], out_dir, args)
    dw, _ = run_pass("write", ["--pmc", "WRITE_SIZE", "--kernel-trace"], out_dir, args)
    fcsv, wcsv = find(df, "counter_collection.csv"), find(dw, "counter_collection.csv")
    f = counter_per_kernel(fcsv, "FETCH_SIZE", SYRK)
    w = counter_per_kernel(wcsv, "WRITE_SIZE", SYRK)
    fetch_b, write_b = 2.0 * sum(f) / len(f) * 1024.0, sum(w) / len(w) * 1024.0
    json.dump({
        "kernel_contains": SYRK,
        "launches": {"fetch_pass": len(f), "write_pass": len(w)},
        "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
        "traffic_bytes_per_launch": fetch_b + write_b,
        "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
        "avg_launch_us": roof["avg_launch_us"], "avg_launch_gflop": roof["avg_launch_gflop"],
        "avg_launch_us_rocprofv3_stats": avg_us_stats,
        "note": "memory-side L2 traffic (Infinity-Cache hits are counted, MI355X_MICROARCH.md); fetch x2 (gfx950 "
                "half-count of wide reads), counters in KiB; avg_launch_us / avg_launch_gflop: what bench.py measured "
                "live with HIP events in the --kernel-trace --stats pass of the same command (rocprofv3's own average "
                "for the kernel in that pass next to it); bench.py flags `traffic_stale` when its live averages move "
                "more than 5 % away from these",
    }, open(os.path.join(prof, f"{tag}_syrk_pmc.json"), "w"), indent=1)

    # HBM-bound kernels: the same two counters on tools/hbm_kernels.py (dense Q assemblies + the solves' GEMV chain)
    hbm = {}
    df, _ = run_pass("hbm_fetch", ["--pmc", "FETCH_SIZE", "--kernel-trace"], out_dir, [], script="tools/hbm_kernels.py")
    dw, line = run_pass("hbm_write", ["--pmc", "WRITE_SIZE", "--kernel-trace"], out_dir, [], script="tools/hbm_kernels.py")
    fcsv, wcsv = find(df, "counter_collection.csv"), find(dw, "counter_collection.csv")
    wq = counter_per_kernel(wcsv, "WRITE_SIZE", "q_assemble_kernel<double")
    tr = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
          for r in rows(find(dw, "kernel_trace.csv")) if "q_assemble_kernel<double" in r["Kernel_Name"]}
    big = sorted(wq)[-5:] if wq else []
    if big:
        n = line["vertices_per_film"]
        ids = [r["Dispatch_Id"] for r in rows(wcsv) if r["Counter_Name"] == "WRITE_SIZE" and "q_assemble_kernel<double" in
               r["Kernel_Name"] and float(r["Counter_Value"]) >= big[0]]
        secs = [tr[i] for i in ids if i in tr]
        hbm["q_assemble_kernel (dense Q, bench extras)"] = {
            "launches": len(big), "WRITE_SIZE_bytes_per_launch": sum(big) / len(big) * 1024.0,
            "algorithmic_bytes": n * n * 8, "avg_ms": sum(secs) / max(1, len(secs)) * 1e3,
            "HBM_write_TBps": (sum(big) / len(big) * 1024.0) / (sum(secs) / max(1, len(secs))) / 1e12 if secs else None}
    fg = [(r["Dispatch_Id"], float(r["Counter_Value"])) for r in rows(fcsv)
          if r["Counter_Name"] == "FETCH_SIZE" and "gemv_batch_kernel<double, 0>" in r["Kernel_Name"]]
    tg = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
          for r in rows(find(df, "kernel_trace.csv")) if "gemv_batch_kernel<double, 0>" in r["Kernel_Name"]}
    if fg:
        tot_b = 2.0 * sum(v for _, v in fg) * 1024.0
        tot_s = sum(tg[i] for i, _ in fg if i in tg)
        hbm["gemv_batch_kernel<double, 0> (the rectangular blocks of the triangular-solve chain, all launches of the run)"] = {
            "launches": len(fg), "FETCH_SIZE_bytes_total_x2": tot_b, "total_ms": tot_s * 1e3,
            "HBM_read_TBps": tot_b / tot_s / 1e12 if tot_s else None}
    hbm["note"] = ("rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of tools/hbm_kernels.py (config H's films: five dense "
                   "Q assemblies, five 11-pass solves; dispatches serialized by the profiler); counters in KiB, FETCH x2 "
                   "on gfx950")
    n_vertices = line["vertices_per_film"]

    dm, _ = run_pass("mfma", ["--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F64",
                              "--kernel-trace"], out_dir, args)
    per = {}
    for r in rows(find(dm, "counter_collection.csv")):
        if SYRK in r["Kernel_Name"]:
            per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
           for r in rows(find(dm, "kernel_trace.csv")) if SYRK in r["Kernel_Name"]}
    sel = [(c, dur[i]) for i, c in per.items() if i in dur and len(c) >= 3]
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c, _ in sel)
    gui = sum(c["GRBM_GUI_ACTIVE"] for c, _ in sel) / XCDS
    mops = sum(c["SQ_INSTS_VALU_MFMA_MOPS_F64"] for c, _ in sel)
    secs = sum(t for _, t in sel)
    json.dump({
        "kernel_contains": SYRK, "launches": len(sel), "avg_launch_us": secs / len(sel) * 1e6,
        "effective_clock_GHz": gui / secs / 1e9, "mfma_busy_fraction_of_active_cycles": busy / (gui * SIMDS),
        "fp64_mfma_flops_per_launch": mops * 512 / len(sel), "achieved_TFLOPs": mops * 512 / secs / 1e12,
        "peak_at_effective_clock_TFLOPs": SIMDS * 32 * (gui / secs) / 1e12,
        "note": "one v_mfma_f64_16x16x4_f64 = 2048 flop per 64 cycles per SIMD -> 32 flop/clk/SIMD; nominal peak 78.6 "
                "TFLOP/s assumes 2.4 GHz; under FP64 MFMA load the chip holds a lower clock (DVFS)",
    }, open(os.path.join(prof, f"{tag}_syrk_mfma_pmc.json"), "w"), indent=1)

    # the all-pairs kernels (inter-film coupling, self field on the rows that are not unknowns): vector-ALU issue and
    # LDS counters on the solves of tools/hbm_kernels.py
    dp, _ = run_pass("pairs", ["--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
                               "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--kernel-trace"],
                     out_dir, [], script="tools/hbm_kernels.py")
    pairs = {}
    trace = {r["Dispatch_Id"]: (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
             for r in rows(find(dp, "kernel_trace.csv"))}
    for needle, label, npairs in (("biot_savart_partial_kernel<double", "biot_savart_partial_kernel<double> (inter-film coupling)", None),
                                  ("self_field_rows_partial_kernel<double", "self_field_rows_partial_kernel<double>", None),
                                  ("q_assemble_kernel<double", "q_assemble_kernel<double> (dense Q: vector-ALU side)", None)):
        per = {}
        for r in rows(find(dp, "counter_collection.csv")):
            if needle in r["Kernel_Name"]:
                per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        sel = [(c, trace[i][1]) for i, c in per.items() if i in trace]
        if not sel:
            continue
        tot = {k: sum(c.get(k, 0.0) for c, _ in sel) for k in sel[0][0]}
        secs = sum(t for _, t in sel)
        gui = tot.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        pairs[label] = {
            "launches": len(sel), "avg_launch_us": secs / len(sel) * 1e6,
            "effective_clock_GHz": gui / secs / 1e9 if secs else None,
            "valu_instructions_per_launch": tot.get("SQ_INSTS_VALU", 0.0) / len(sel),
            "valu_active_quad_cycles_per_launch": tot.get("SQ_ACTIVE_INST_VALU", 0.0) / len(sel),
            "wave_quad_cycles_per_launch": tot.get("SQ_WAVE_CYCLES", 0.0) / len(sel),
            "valu_active_fraction_of_wave_cycles": (tot.get("SQ_ACTIVE_INST_VALU", 0.0) / tot["SQ_WAVE_CYCLES"]
                                                    if tot.get("SQ_WAVE_CYCLES") else None),
            "lds_instructions_per_launch": tot.get("SQ_INSTS_LDS", 0.0) / len(sel),
            "lds_bank_conflict_cycles_per_launch": tot.get("SQ_LDS_BANK_CONFLICT", 0.0) / len(sel),
            "lds_active_cycles_per_launch": tot.get("SQ_LDS_IDX_ACTIVE", 0.0) / len(sel),
        }
    # the dense Q assembly's issue side goes with its memory side (hbm_pmc): is it the FP64 work or the store stream?
    qk = "q_assemble_kernel<double> (dense Q: vector-ALU side)"
    if qk in pairs:
        q = pairs.pop(qk)
        # the five dense launches are the long ones of the pass (the row-sum-only launches of the factorizations are short)
        q["valu_instructions_per_output_element"] = q["valu_instructions_per_launch"] * 64.0 / float(n_vertices) ** 2
        q["note"] = ("all q_assemble launches of the pass (dense Q and the factorizations' row-sum-only launches alike); "
                     "SQ_INSTS_VALU counts wave instructions: x 64 lanes / n^2 elements")
        hbm["q_assemble_kernel (vector-ALU counters)"] = q
    json.dump(hbm, open(os.path.join(prof, f"{tag}_hbm_pmc.json"), "w"), indent=1)
    pairs["note"] = ("rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                     "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass of tools/hbm_kernels.py (config H's films, five "
                     "11-pass solves: 25 117 x ~22 000 pairs per coupling launch); SQ_* activity counters are quad-cycles")
    json.dump(pairs, open(os.path.join(prof, f"{tag}_biot_savart_pmc.json"), "w"), indent=1)

    p = subprocess.run(["python3", os.path.join(ROOT, "bench.py")], cwd=ROOT, capture_output=True, text=True)
    full = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if full:
        json.dump(json.loads(full[-1]), open(os.path.join(prof, f"{tag}_bench_full.json"), "w"), indent=1)
    print("summaries in", prof, os.listdir(prof))


if __name__ == "__main__":
    main()
