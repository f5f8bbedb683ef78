"""SYRK launches of the last factorization in a rocprofv3 kernel trace: order, tiles, duration (development aid).
usage: python tools/syrk_in_trace.py <trace dir> [films]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
asm = [i for i, r in enumerate(rows) if "system_assemble" in r["Kernel_Name"]]
rows = rows[asm[-nf]:]
syrk = [r for r in rows if "gemm_op_kernel" in r["Kernel_Name"] and ", true>" in r["Kernel_Name"]]
t0 = rows[0]["s"]
print(f"{len(syrk)} SYRK launches")
for k, r in enumerate(syrk):
    if k < 16 or k % 16 == 0:
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))); wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)))
        tiles = gx // max(wg, 1)
        nt = int(((8 * tiles + 1) ** 0.5 - 1) / 2)
        M = 128 * nt
        us = (r["e"] - r["s"]) / 1e3
        # flops for K = 256 and 512
        print(f"#{k:3d} @ {(r['s'] - t0) / 1e6:7.1f} ms  tiles {tiles:6d}  M ~ {M:6d}  {us:8.1f} us   TFLOP/s if K=256: {M * (M + 128) * 256 / us / 1e6:5.1f}   if K=512: {M * (M + 128) * 512 / us / 1e6:5.1f}")
