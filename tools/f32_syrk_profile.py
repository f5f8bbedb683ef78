"""The float32 trailing update (SYRK) of config H's factorization, profiled on its own:

  pass 1  rocprofv3 --kernel-trace                     -> per-launch microseconds in situ, against the same launch ALONE
  pass 2  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
                                                       -> MFMA pipes busy / active cycles, effective clock

of `python3 tools/fact_timing.py auto float32` (7 cold factorizations; the last one is analysed).  Run on the GPU box
from the repo root:   python tools/f32_syrk_profile.py [out.txt]
Counters in their own pass, nothing but --kernel-trace beside them (MI355X_MICROARCH.md).
"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NEEDLE = "gemm_op_kernel<float, 0, 1, true>"
SIMDS, XCDS = 256 * 4, 8
PEAK_F32 = 157.3   # TFLOP/s, dense FP32 matrix (v_mfma_f32_16x16x4_f32: 64 flop / clk / SIMD at 2.4 GHz)


def rocprof(name, flags):
    d = os.path.join("/tmp", "ssa_f32_" + name)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3"] + flags + ["--output-format", "csv", "-d", d, "-o", name, "--", "python3",
           os.path.join(ROOT, "tools", "fact_timing.py"), "auto", "float32"]
    p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout[-2000:] + p.stderr[-3000:])
        raise SystemExit(f"pass {name} failed")
    line = [ln for ln in p.stdout.splitlines() if "factorize median" in ln]
    return d, (line[-1] if line else "")


def find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True))[-1]


def last_factorization(trace_csv):
    rows = list(csv.DictReader(open(trace_csv)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    asm = [i for i, r in enumerate(rows) if "system_assemble" in r["Kernel_Name"]]
    return rows[asm[-2]:]      # two films: the last two assemblies open the last factorization


def schedule(unknowns, tail=10240):
    """(film, M, K) of every stand-alone trailing update, in launch order (chol.hip: potrf_batch, two or more matrices)."""
    out = []
    npads = [-(-n // 256) * 256 for n in unknowns]
    nmax = max(npads)
    upd0 = [0] * len(npads)
    for k0 in range(0, nmax - 256, 256):
        c = k0 + 256
        if nmax - c <= tail:
            break
        for f, npad in enumerate(npads):
            if c >= npad:
                continue
            right, kp = npad - c, c - upd0[f]
            delay = kp < 512 and right > 8192 and ((k0 + npad) // 256) % 2 != 1
            if right > 256 and not delay:
                out.append((f, npad - (c + 256), kp))
            if not delay:
                upd0[f] = c
    return out


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout

    def say(*a):
        print(*a, file=out, flush=True)

    import torch

    from superscreen_amd import kernels as K

    # both profiler passes first: they are child processes, and a process that has initialised the GPU must not start
    # other programs on this pool
    d1, line1 = rocprof("trace", ["--kernel-trace"])
    d2, _ = rocprof("mfma", ["--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--kernel-trace"])
    rows = last_factorization(find(d1, "kernel_trace.csv"))
    syrk = [r for r in rows if NEEDLE in r["Kernel_Name"]]
    t0 = rows[0]["s"]
    t1 = max(r["e"] for r in rows if "chol_tail_round" in r["Kernel_Name"] or "transpose_lower" in r["Kernel_Name"])
    say(f"# {line1}")
    say(f"# last factorization of the traced run: {(t1 - t0) / 1e6:.2f} ms, {len(syrk)} float32 SYRK launches")
    Mmax = 0
    launches = []
    sched = schedule([18150, 20419])
    say(f"# schedule model: {len(sched)} launches" + ("" if len(sched) == len(syrk) else " (differs from the trace: K taken as 512)"))
    for k, r in enumerate(syrk):
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)))
        tiles = gx // max(wg, 1)
        nt = int(((8 * tiles + 1) ** 0.5 - 1) / 2)
        M = 128 * nt
        Kd = sched[k][2] if len(sched) == len(syrk) and sched[k][1] == M else 512
        launches.append((r, M, Kd))
        Mmax = max(Mmax, M)
    Cbuf = torch.randn((Mmax, Mmax), dtype=torch.float32, device="cuda")
    alone = {}

    def alone_us(M, Kd):
        if (M, Kd) not in alone:
            P = torch.randn((M, Kd), dtype=torch.float32, device="cuda")
            C = Cbuf[:M]
            for _ in range(2):
                K.gemm_ex(0, 1, True, P, P, C, M, M, Kd, alpha=-1e-3, beta=1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                K.gemm_ex(0, 1, True, P, P, C, M, M, Kd, alpha=-1e-3, beta=1.0)
            e1.record()
            torch.cuda.synchronize()
            alone[(M, Kd)] = e0.elapsed_time(e1) / 6 * 1e3
        return alone[(M, Kd)]

    say(f"{'#':>3} {'start ms':>9} {'M':>6} {'K':>4} {'in situ us':>11} {'alone us':>9} {'ratio':>6} {'TFLOP/s in situ':>16} {'alone':>6}")
    tin = tal = fl = 0.0
    for k, (r, M, Kd) in enumerate(launches):
        us, al, f = (r["e"] - r["s"]) / 1e3, alone_us(M, Kd), Kd * M * (M + 128)
        tin, tal, fl = tin + us, tal + al, fl + f
        say(f"{k:3d} {(r['s'] - t0) / 1e6:9.2f} {M:6d} {Kd:4d} {us:11.1f} {al:9.1f} {us / al:6.3f} {f / us / 1e6:16.1f} {f / al / 1e6:6.1f}")
    say(f"sum: in situ {tin / 1e3:.2f} ms = {fl / tin / 1e6:.1f} TFLOP/s = {fl / tin / 1e6 / PEAK_F32:.3f} of the FP32 matrix peak; "
        f"alone {tal / 1e3:.2f} ms = {fl / tal / 1e6:.1f} TFLOP/s = {fl / tal / 1e6 / PEAK_F32:.3f}; ratio {tin / tal:.3f}")
    say(f"the rest of the factorization (chains not hidden, rounds, finishing): {(t1 - t0) / 1e6 - tin / 1e3:.2f} ms of {(t1 - t0) / 1e6:.2f}")

    per = {}
    for r in csv.DictReader(open(find(d2, "counter_collection.csv"))):
        if NEEDLE in r["Kernel_Name"]:
            per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
           for r in csv.DictReader(open(find(d2, "kernel_trace.csv"))) if NEEDLE in r["Kernel_Name"]}
    sel = [(c, dur[i]) for i, c in per.items() if i in dur and len(c) >= 2]
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c, _ in sel)
    gui = sum(c["GRBM_GUI_ACTIVE"] for c, _ in sel) / XCDS
    secs = sum(t for _, t in sel)
    say(f"counter pass: {len(sel)} float32 SYRK launches, avg {secs / max(1, len(sel)) * 1e6:.1f} us; MFMA pipes busy "
        f"{busy / (gui * SIMDS):.3f} of the active cycles; effective clock {gui / secs / 1e9:.2f} GHz "
        f"(peak at that clock: {SIMDS * 64 * gui / secs / 1e12:.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
