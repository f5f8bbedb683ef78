"""Timeline of the single-stream rounds of a Cholesky factorization from a rocprofv3 kernel trace (development aid).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rt -- python3 tools/r04/round_timeline.py run [float64|stack4|single]
    python tools/r04/round_timeline.py analyse gpurun_out/rt [nfilms]"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(kind):
    import torch

    sys.path.insert(0, ROOT)
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    if kind == "stack4":
        device = synthetic.make_stack_device(100, ("disk",) * 4, solve_dtype="float64")
    elif kind == "single":
        device = synthetic.make_stack_device(129, ("disk",), solve_dtype="float64")
    else:
        device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype=kind)
    for _ in range(3):
        model = sc.factorize_model(device=device, current_units="uA")
        torch.cuda.synchronize()
        del model


def short(name):
    for k, v in (("chol_tail_round", "ROUND"), ("gemm_nt_small_batch", "small_batch"), ("chol_diag256", "diag"),
                 ("Lb1EEE", "SYRK"), ("gemm_op_kernel", "gemm_op"), ("gemm_nt_small", "small"), ("gemm_kernel", "gemm_nn"),
                 ("transpose_lower", "transpose"), ("system_assemble", "assemble"), ("q_assemble", "q_rowsum"),
                 ("fillBuffer", "fill")):
        if k in name:
            return v
    return name[:24]


def analyse(d, nf):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["k"] = short(r["Kernel_Name"])
        if r["k"] == "gemm_op" and ("true>" in r["Kernel_Name"] or "Lb1" in r["Kernel_Name"]):
            r["k"] = "SYRK"
    rows.sort(key=lambda r: r["s"])
    asm = [i for i, r in enumerate(rows) if r["k"] == "assemble"]
    rows = rows[asm[-nf]:]
    t0, t1 = rows[0]["s"], max(r["e"] for r in rows)
    print(f"factorization span {(t1 - t0) / 1e6:.2f} ms, {len(rows)} launches")
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        tot[r["k"]][0] += 1
        tot[r["k"]][1] += (r["e"] - r["s"]) / 1e6
    for k, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:14s} {c:5d} launches {ms:8.2f} ms summed")
    rounds = [i for i, r in enumerate(rows) if r["k"] == "ROUND"]
    if not rounds:
        return
    print(f"first round launch at {(rows[rounds[0]]['s'] - t0) / 1e6:.2f} ms; {len(rounds)} rounds; last ends {(rows[rounds[-1]]['e'] - t0) / 1e6:.2f} ms")
    q = rows[rounds[0]]["Queue_Id"]
    for n, i in enumerate(rounds):
        j = rounds[n + 1] if n + 1 < len(rounds) else len(rows)
        seq = [r for r in rows[i:j] if r["Queue_Id"] == q]
        others = [r for r in rows[i:j] if r["Queue_Id"] != q]
        parts, prev = [], None
        for r in seq:
            gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
            parts.append(f"{'+%.0f ' % gap if prev is not None else ''}{r['k']} {(r['e'] - r['s']) / 1e3:.0f} [{r.get('Grid_Size', '?')}]")
            prev = r["e"]
        nxt = rows[j]["s"] if j < len(rows) else t1
        oth = collections.Counter(r["k"] for r in others)
        print(f"round {n:3d} @ {(rows[i]['s'] - t0) / 1e6:6.2f} ms  {(nxt - rows[i]['s']) / 1e3:6.0f} us : " + " | ".join(parts) +
              (f"   beside: {dict(oth)}" if oth else ""))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "float64")
    else:
        analyse(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2)
