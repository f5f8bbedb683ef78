"""Per-launch table of the trailing-update (SYRK) launches of one factorization: M, K, microseconds inside the
factorization (from a rocprofv3 kernel trace of bench.py) and microseconds of the same launch ALONE on an idle
device (measured here), so that the loss to what shares the matrix pipes is attributed launch by launch.
usage: python tools/r04/syrk_launch_table.py <trace dir> [unknowns ...]      (default 18150 20419: config H)"""
import csv, glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from superscreen_amd import kernels as K

trace = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
unknowns = [int(a) for a in sys.argv[2:]] or [18150, 20419]
rows = list(csv.DictReader(open(trace)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
asm = [i for i, r in enumerate(rows) if "system_assemble" in r["Kernel_Name"]]
first = asm[-len(unknowns)]
rows = rows[first:]
t0 = rows[0]["s"]
syrk = [r for r in rows if "gemm_op_kernel<double" in r["Kernel_Name"] and ", true>" in r["Kernel_Name"]]
last_fact = max(r["e"] for r in rows if "chol_tail_round" in r["Kernel_Name"] or r in syrk)


def schedule(unknowns):
    """(film, M, K) of every stand-alone update launch, in launch order (chol.hip potrf_batch; bench.py chol_schedule)"""
    out = []
    npads = [-(-n // 256) * 256 for n in unknowns]
    nmax = max(npads)
    upd0 = [0] * len(npads)
    for k0 in range(0, nmax - 256, 256):
        c = k0 + 256
        if nmax - c <= 10240:
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


sched = schedule(unknowns)
print(f"# {trace}")
print(f"# {len(syrk)} SYRK launches in the trace, {len(sched)} in the schedule model; factorization span "
      f"{(last_fact - t0) / 1e6:.2f} ms")
if len(sched) != len(syrk):
    print("# the model and the trace disagree: K taken as 512")
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
      f"{'TFLOP/s in situ':>16} {'alone':>6}")
tot_in = tot_al = flops = 0.0
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
    print(f"{k:3d} {(r['s'] - t0) / 1e6:9.2f} {f:4d} {M:6d} {Kd:4d} {us:11.1f} {al:9.1f} {us / al:6.3f} "
          f"{fl / us / 1e6:16.1f} {fl / al / 1e6:6.1f}")
print(f"sum: in situ {tot_in / 1e3:.2f} ms ({flops / tot_in / 1e6:.1f} TFLOP/s), alone {tot_al / 1e3:.2f} ms "
      f"({flops / tot_al / 1e6:.1f} TFLOP/s), ratio {tot_in / tot_al:.3f}")
