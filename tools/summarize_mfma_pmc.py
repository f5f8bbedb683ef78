"""MFMA-pipe utilisation and effective clock of one kernel from a rocprofv3 PMC pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace \
              --output-format csv -d <out> -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/summarize_mfma_pmc.py <out>/m_counter_collection.csv <out>/m_kernel_trace.csv \
              "gemm_op_kernel<double, 0, 1, true>" profiles/r01_v5_syrk_mfma_pmc.json

MfmaUtil (counter_defs.yaml, gfx950) = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM).
rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note): the per-XCD
active cycles are the value / 8, and the effective shader clock is that quotient / kernel duration.
SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 = FP64 MFMA flops executed.
"""
import csv
import json
import sys

SIMDS = 256 * 4
XCDS = 8


def main():
    counters_csv, trace_csv, needle, out = sys.argv[1:5]
    per_dispatch = {}
    for r in csv.DictReader(open(counters_csv)):
        if needle in r["Kernel_Name"]:
            per_dispatch.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    durations = {}
    for r in csv.DictReader(open(trace_csv)):
        if needle in r["Kernel_Name"]:
            durations[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    rows = [(c, durations[d]) for d, c in per_dispatch.items() if d in durations and len(c) >= 3]
    n = len(rows)
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c, _ in rows)
    gui = sum(c["GRBM_GUI_ACTIVE"] for c, _ in rows) / XCDS
    mops = sum(c["SQ_INSTS_VALU_MFMA_MOPS_F64"] for c, _ in rows)
    secs = sum(t for _, t in rows)
    clock = gui / secs
    res = {
        "kernel_contains": needle,
        "launches": n,
        "avg_launch_us": secs / n * 1e6,
        "effective_clock_GHz": clock / 1e9,
        "mfma_busy_fraction_of_active_cycles": busy / (gui * SIMDS),
        "fp64_mfma_flops_per_launch": mops * 512 / n,
        "achieved_TFLOPs": mops * 512 / secs / 1e12,
        "peak_at_effective_clock_TFLOPs": SIMDS * 32 * clock / 1e12,
        "frac_of_peak_at_effective_clock": (mops * 512 / secs) / (SIMDS * 32 * clock),
        "note": "one v_mfma_f64_16x16x4_f64 = 2048 flop per 64 cycles per SIMD -> 32 flop/clk/SIMD; nominal peak "
                "78.6 TFLOP/s assumes 2.4 GHz; under FP64 MFMA load the chip holds a lower clock (DVFS)",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
