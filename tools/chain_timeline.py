"""Chain-stream timeline of one Cholesky factorization from a rocprofv3 kernel trace (development aid):
for every chol_diag256 launch of the busiest chain queue, the kernels between it and the next one.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -- python3 tools/chain_timeline.py run float64
    python tools/chain_timeline.py analyse /tmp/ct"""
import collections
import csv
import glob
import os
import sys


def run(dtype):
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    if dtype == "stack4":   # the 4-film stack of config 5 (float64)
        device = synthetic.make_stack_device(100, ("disk",) * 4, solve_dtype="float64")
    else:
        device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype=dtype)
    for _ in range(2):
        model = sc.factorize_model(device=device, current_units="uA")
        torch.cuda.synchronize()
        del model


def analyse(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    asm = [i for i, r in enumerate(rows) if "system_assemble" in r["Kernel_Name"]]
    nf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = rows[asm[-nf]:]         # the last factorization (one assembly per film)
    t0 = rows[0]["s"]
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r["Queue_Id"]].append(r)
    chains = [(q, rs) for q, rs in byq.items() if any("chol_diag256" in r["Kernel_Name"] for r in rs)]
    chains.sort(key=lambda kv: -len(kv[1]))
    short = {"chol_diag256": "diag", "gemm_op_kernel": "gemm", "gemm_nt_small": "small"}
    for q, rs in chains:
      diag = [i for i, r in enumerate(rs) if "chol_diag256" in r["Kernel_Name"]]
      print(f"chain queue {q}: {len(diag)} panels; factorization span {(rows[-1]['e'] - t0) / 1e6:.1f} ms")
      for n, i in enumerate(diag[:-1]):
        if not (n % 8 == 0 or n >= len(diag) - 6):
            continue
        j = diag[n + 1]
        parts = []
        for r in rs[i:j]:
            nm = next((v for k, v in short.items() if k in r["Kernel_Name"]), r["Kernel_Name"][:12])
            gap = (r["s"] - prev_e) / 1e3 if parts else 0.0
            parts.append(f"{'+%.0f ' % gap if parts else ''}{nm} {(r['e'] - r['s']) / 1e3:.0f}")
            prev_e = r["e"]
        print(f"panel {n:3d} @ {(rs[i]['s'] - t0) / 1e6:6.1f} ms  round {(rs[j]['s'] - rs[i]['s']) / 1e3:6.0f} us :  " + " | ".join(parts))
    syrk = [r for r in rows if "Lb1EEE" in r["Kernel_Name"] or ", true>" in r["Kernel_Name"] and "gemm_op" in r["Kernel_Name"]]
    if syrk:
        busy = sum(r["e"] - r["s"] for r in syrk)
        print(f"SYRK launches {len(syrk)}, busy {busy / 1e6:.1f} ms, first {(syrk[0]['s'] - t0) / 1e6:.1f}, last end {(syrk[-1]['e'] - t0) / 1e6:.1f} ms")
        span = rows[-1]["e"] - t0
        win = 5_000_000 if span < 150_000_000 else 20_000_000
        for uq in sorted({r["Queue_Id"] for r in syrk}):
            line = []
            for w0 in range(0, span, win):
                b = sum(max(0, min(r["e"] - t0, w0 + win) - max(r["s"] - t0, w0)) for r in byq[uq])
                line.append(f"{100 * b / min(win, span - w0):.0f}")
            print(f"update queue {uq} busy % per {win // 1_000_000} ms window: " + " ".join(line))
        # all SYRK launches together: how much of the time is at least one / how many run at once
        ev = sorted([(r["s"], 1) for r in syrk] + [(r["e"], -1) for r in syrk])
        depth, last, hist = 0, ev[0][0], collections.Counter()
        for t, d in ev:
            hist[depth] += t - last
            depth, last = depth + d, t
        print("SYRK launches in flight (ms): " + ", ".join(f"{k}: {v / 1e6:.1f}" for k, v in sorted(hist.items())))
        uq = syrk[0]["Queue_Id"]
        other = collections.defaultdict(lambda: [0, 0, 0])
        for r in byq[uq]:
            if r in syrk:
                continue
            o = other[r["Kernel_Name"][:70]]
            o[0] += 1
            o[1] += r["e"] - r["s"]
            o[2] += (r["e"] - r["s"]) if r["s"] >= syrk[-1]["e"] else 0
        for k, o in sorted(other.items(), key=lambda kv: -kv[1][1]):
            print(f"  update queue, not SYRK: {o[0]:4d} x {k:70s} {o[1] / 1e6:6.2f} ms ({o[2] / 1e6:.2f} ms after the last SYRK)")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        analyse(sys.argv[2])
