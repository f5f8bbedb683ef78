"""What runs around and after the last round of a factorization: every kernel that starts in the last 3 ms of the
last factorization of a kernel trace (tools/r04/round_timeline.py run ...), with its queue, start, duration and
grid (development aid).   usage: python tools/r04/tail_listing.py <trace dir> [ms]"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from round_timeline import short
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rounds = [r for r in rows if "chol_tail_round" in r["Kernel_Name"]]
last = rounds[-1]
asm = [r for r in rows if "system_assemble" in r["Kernel_Name"] and r["s"] < last["s"]][-1]
t_end = max(r["e"] for r in rows if r["s"] >= last["s"] and r["s"] < last["e"] + 5_000_000 and
            any(k in r["Kernel_Name"] for k in ("gemm", "transpose", "chol", "mirror")))
print(f"last round launch starts {(last['s'] - asm['s']) / 1e6:.3f} ms after the last assembly, ends +{(last['e'] - last['s']) / 1e3:.0f} us; "
      f"last factorization kernel ends {(t_end - last['e']) / 1e3:.0f} us after it")
for r in rows:
    if r["s"] < last["s"] - span * 1e6 or r["s"] > t_end:
        continue
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256))))
    print(f"{(r['s'] - last['s']) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {(r['e'] - r['s']) / 1e3:7.1f} us  wgs {gx:6d}  {short(r['Kernel_Name'])}")
