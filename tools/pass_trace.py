"""Kernel trace helper (development aid): one factorization, then three warm 10-iteration solves of config H.
    rocprofv3 --kernel-trace --output-format csv -d <out> -- python3 tools/pass_trace.py
    python tools/pass_trace.py analyse <out>"""
import collections, csv, glob, os, sys

def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import superscreen_amd as sc
    from superscreen_amd import synthetic
    device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
    model = sc.factorize_model(device=device, current_units="uA")
    for i in range(3):
        sc.solve(model=model, applied_field=sc.ConstantField(0.5 + i), iterations=10, progress_bar=False)
    torch.cuda.synchronize()

def analyse(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    last = max(i for i, r in enumerate(rows) if "chol_diag" in r["Kernel_Name"])
    rows = rows[last + 1:]
    # skip the finishing passes of the factorization: start at the first film_rhs-like kernel
    start = next(i for i, r in enumerate(rows) if "film_rhs" in r["Kernel_Name"] or "rhs" in r["Kernel_Name"].lower())
    rows = rows[start:]
    span = (rows[-1]["e"] - rows[0]["s"]) / 1e6
    busy = sum(r["e"] - r["s"] for r in rows) / 1e6
    print(f"3 warm solves: span {span:.2f} ms, summed kernel time {busy:.2f} ms, {len(rows)} launches")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][-70:]
        agg[k][0] += 1
        agg[k][1] += (r["e"] - r["s"]) / 1e3
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{us / 3e3:8.3f} ms/solve  n/solve {n / 3:6.1f}  avg {us / n:7.1f} us  {k}")
    gaps = sum(max(0, b["s"] - a["e"]) for a, b in zip(rows, rows[1:])) / 1e6
    print(f"gaps between consecutive kernels: {gaps:.2f} ms over 3 solves")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "analyse":
        analyse(sys.argv[2])
    else:
        run()
