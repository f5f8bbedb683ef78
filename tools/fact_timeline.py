"""Where the time of one Cholesky factorization goes, from a rocprofv3 kernel trace (development aid; replaces
tools/r04/syrk_launch_table.py and tools/r04/round_timeline.py):

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ft -- python3 tools/fact_timeline.py run [float64|float32|stack4|single]
    python tools/fact_timeline.py table   gpurun_out/ft [unknowns ...]   per trailing-update (SYRK) launch: in situ / alone / BESIDE
    python tools/fact_timeline.py rounds  gpurun_out/ft [nfilms]         per round: launches on the round's queue, gaps, what ran beside

`table` needs the GPU (it times every launch shape alone on the idle device); `beside` lists, per SYRK launch, the
kernels of OTHER queues that overlapped it with the microseconds of overlap (summed per kernel kind), so that a
launch that takes 1.9 x its time alone is explained by what shared the chip with it."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    if "gemm_op_kernel" in name and (", true>" in name or "Lb1EEE" in name):
        return "SYRK"
    for k, v in (("chol_tail_round", "ROUND"), ("chol_fused_round", "ROUND"), ("gemm_nt_small_batch", "small_batch"),
                 ("chol_diag256", "diag"), ("gemm_op_kernel", "strip/panel"), ("gemm_nt_small", "small"),
                 ("gemm_slice", "inv_slice"), ("gemm_kernel", "inv_gemm"), ("transpose_lower", "transpose"),
                 ("system_assemble", "assemble"), ("q_assemble", "q_rowsum"), ("fillBuffer", "fill")):
        if k in name:
            return v
    return name[:24]


def load(d, nf):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["k"] = short(r["Kernel_Name"])
    rows.sort(key=lambda r: r["s"])
    asm = [i for i, r in enumerate(rows) if r["k"] == "assemble"]
    rows = rows[asm[-nf]:]          # the last factorization of the trace
    return f, rows


def schedule(unknowns, tail_cols=10240):
    """(film, M, K) of every stand-alone update launch, in launch order (chol.hip potrf_batch; bench.py chol_schedule)"""
    out = []
    npads = [-(-n // 256) * 256 for n in unknowns]
    nmax = max(npads)
    upd0 = [0] * len(npads)
    for k0 in range(0, nmax - 256, 256):
        c = k0 + 256
        if nmax - c <= tail_cols:
            break
        for f, npad in enumerate(npads):
            if c >= npad:
                continue
            right = npad - c
            kp = c - upd0[f]
            delay = kp < 512 and right > 8192 and ((k0 + npad) // 256) % 2 != 1
            if right > 256 and not delay:
                out.append((f, npad - (c + 256), kp))
            if not delay:
                upd0[f] = c
    return out


def table(d, unknowns):
    import torch

    sys.path.insert(0, ROOT)
    from superscreen_amd import kernels as K

    trace, rows = load(d, len(unknowns))
    t0 = rows[0]["s"]
    syrk = [r for r in rows if r["k"] == "SYRK"]
    last_fact = max(r["e"] for r in rows if r["k"] in ("ROUND", "DAG", "SYRK"))
    sched = schedule(unknowns)
    print(f"# {trace}")
    print(f"# {len(syrk)} SYRK launches in the trace, {len(sched)} in the schedule model; factorization span "
          f"{(last_fact - t0) / 1e6:.2f} ms")
    alone = {}
    Mmax = max(m for _, m, _ in sched)
    Cbuf = torch.randn((Mmax, Mmax), dtype=torch.float64, device="cuda")

    def alone_us(M, Kd):
        if (M, Kd) not in alone:
            P = torch.randn((M, Kd), dtype=torch.float64, device="cuda")
            C = Cbuf[:M]
            for _ in range(2):
                K.gemm_ex(0, 1, True, P, P, C, M, M, Kd, alpha=-1e-3, beta=1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 6
            e0.record()
            for _ in range(reps):
                K.gemm_ex(0, 1, True, P, P, C, M, M, Kd, alpha=-1e-3, beta=1.0)
            e1.record()
            torch.cuda.synchronize()
            alone[(M, Kd)] = e0.elapsed_time(e1) / reps * 1e3
        return alone[(M, Kd)]

    print(f"{'#':>3} {'start ms':>9} {'film':>4} {'M':>6} {'K':>4} {'in situ us':>11} {'alone us':>9} {'ratio':>6} "
          f"{'TF in situ':>10} {'alone':>6}  beside (us of overlap per kind, other queues)")
    tot_in = tot_al = flops = 0.0
    worst = 0.0
    besides = collections.Counter()
    for k, r in enumerate(syrk):
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)))
        tiles = gx // max(wg, 1)
        nt = int(((8 * tiles + 1) ** 0.5 - 1) / 2)
        M = 128 * nt
        f, Ms, Kd = sched[k] if len(sched) == len(syrk) else (-1, M, 512)
        if Ms != M:
            f, Kd = -1, 512
        us = (r["e"] - r["s"]) / 1e3
        al = alone_us(M, Kd)
        fl = Kd * M * (M + 128)
        tot_in, tot_al, flops = tot_in + us, tot_al + al, flops + fl
        worst = max(worst, us / al)
        ov = collections.Counter()
        for o in rows:
            if o is r or o["e"] <= r["s"] or o["s"] >= r["e"]:
                continue
            ov[o["k"]] += (min(o["e"], r["e"]) - max(o["s"], r["s"])) / 1e3
        besides.update(ov)
        txt = "  ".join(f"{kk} {v:.0f}" for kk, v in sorted(ov.items(), key=lambda kv: -kv[1]))
        print(f"{k:3d} {(r['s'] - t0) / 1e6:9.2f} {f:4d} {M:6d} {Kd:4d} {us:11.1f} {al:9.1f} {us / al:6.3f} "
              f"{fl / us / 1e6:10.1f} {fl / al / 1e6:6.1f}  {txt}")
    print(f"sum: in situ {tot_in / 1e3:.2f} ms ({flops / tot_in / 1e6:.1f} TFLOP/s), alone {tot_al / 1e3:.2f} ms "
          f"({flops / tot_al / 1e6:.1f} TFLOP/s), ratio {tot_in / tot_al:.3f}, worst launch {worst:.3f}")
    print("overlap with the SYRK launches, summed (ms): " +
          "  ".join(f"{kk} {v / 1e3:.2f}" for kk, v in sorted(besides.items(), key=lambda kv: -kv[1])))


def rounds(d, nf):
    trace, rows = load(d, nf)
    t0, t1 = rows[0]["s"], max(r["e"] for r in rows)
    print(f"# {trace}")
    print(f"factorization span {(t1 - t0) / 1e6:.2f} ms, {len(rows)} launches")
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        tot[r["k"]][0] += 1
        tot[r["k"]][1] += (r["e"] - r["s"]) / 1e6
    for k, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:14s} {c:5d} launches {ms:8.2f} ms summed")
    rnd = [i for i, r in enumerate(rows) if r["k"] in ("ROUND", "DAG")]
    if not rnd:
        return
    print(f"first round launch at {(rows[rnd[0]]['s'] - t0) / 1e6:.2f} ms; {len(rnd)} round launches; last ends "
          f"{(rows[rnd[-1]]['e'] - t0) / 1e6:.2f} ms  -> rounds {(rows[rnd[-1]]['e'] - rows[rnd[0]]['s']) / 1e6:.2f} ms, "
          f"after the last round {(t1 - rows[rnd[-1]]['e']) / 1e6:.2f} ms")
    q = rows[rnd[0]]["Queue_Id"]
    for n, i in enumerate(rnd):
        j = rnd[n + 1] if n + 1 < len(rnd) else len(rows)
        seq = [r for r in rows[i:j] if r["Queue_Id"] == q]
        others = [r for r in rows[i:j] if r["Queue_Id"] != q]
        parts, prev = [], None
        for r in seq:
            gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
            parts.append(f"{'+%.0f ' % gap if prev is not None else ''}{r['k']} {(r['e'] - r['s']) / 1e3:.0f}")
            prev = r["e"]
        nxt = rows[j]["s"] if j < len(rows) else t1
        oth = collections.Counter(r["k"] for r in others)
        print(f"round {n:3d} @ {(rows[i]['s'] - t0) / 1e6:6.2f} ms  {(nxt - rows[i]['s']) / 1e3:6.0f} us : " + " | ".join(parts) +
              (f"   beside: {dict(oth)}" if oth else ""))


def head(d, nf):
    """Every kernel from the first row-sum kernel of the last factorization to its first trailing update."""
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["k"] = short(r["Kernel_Name"])
    rows.sort(key=lambda r: r["s"])
    asm = [i for i, r in enumerate(rows) if r["k"] == "assemble"]
    first_asm = asm[-nf]
    start = first_asm
    while start > 0 and rows[first_asm]["s"] - rows[start - 1]["s"] < 3_000_000 and rows[start - 1]["k"] != "transpose":
        start -= 1
    t0 = rows[start]["s"]
    for r in rows[start:]:
        print(f"{(r['s'] - t0) / 1e3:9.1f} us  +{(r['e'] - r['s']) / 1e3:8.1f} us  queue {r['Queue_Id']:>3}  {r['k']}")
        if r["k"] == "SYRK":
            break


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "float64")
    elif mode == "head":
        head(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2)
    elif mode == "table":
        table(sys.argv[2], [int(a) for a in sys.argv[3:]] or [18150, 20419])
    else:
        rounds(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2)
