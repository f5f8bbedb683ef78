"""Cold factorization + self-consistent solve of the benchmark devices over and over; every film's stream function of
every iterate must come out BIT-IDENTICAL each time (a factorization that differs in the 10th digit shows here; the
schedules use side streams, events and hand-rolled barriers: tools/chol_race_hunt.py locates a difference, this tool
shows whether there is one), and the residual of the film systems is checked on the first repetitions.

    python tools/stress_factorization.py [reps=30] [case ...]

Cases (default: all): H (config H, 2 x 25 117), H32 (float32), Hlu (LU route), c3 (config 3, K = 81), c5 (config 5:
four disks, K = 100 -- the only case whose trailing updates run on streams of their own before the rounds), s4 (four
small films, K = 60: rounds from the first panel on), mix (routes and precisions interleaved in one process).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import poison  # noqa: E402

poison.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

import superscreen_amd as sc  # noqa: E402
from superscreen_amd import kernels, synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
CASES = {
    "H": (91, ("washer", "disk"), 0.5, "float64", "auto", 3),
    "H32": (91, ("washer", "disk"), 0.5, "float32", "auto", 3),
    "Hlu": (91, ("washer", "disk"), 0.5, "float64", "lu", 3),
    "c3": (81, ("washer", "disk"), 0.5, "float64", "auto", 3),
    "c5": (100, ("disk",) * 4, 0.5, "float64", "auto", 2),
    "s4": (60, ("washer", "disk", "washer", "disk"), 1.5, "float64", "auto", 3),
}
wanted = sys.argv[2:] or list(CASES) + ["mix"]
failed = 0

# SSA_STRESS_DISTURB=1: a second stream keeps the chip's LDS and memory pipes busy with transposes of an 8192 x 8192
# matrix all the time (what exposed the in-flight LDS reads of tile_small_nt was company of exactly that kind: the
# finishing passes' transposes beside the rounds).  Results must not depend on it.
if os.environ.get("SSA_STRESS_DISTURB"):
    import threading

    def disturb():
        torch.cuda.set_device(0)
        side = torch.cuda.Stream()
        x = torch.randn(8192, 8192, dtype=torch.float64, device="cuda")
        evs = []
        while True:
            with torch.cuda.stream(side):
                y = x.t().contiguous()   # noqa: F841
                e = torch.cuda.Event()
                e.record(side)
            evs.append(e)
            if len(evs) > 3:
                evs.pop(0).synchronize()

    threading.Thread(target=disturb, daemon=True).start()
    print("disturber running", flush=True)


def residuals(model):
    worst = 0.0
    for name, system in model.film_systems.items():
        if system.chol is None:
            continue
        fd = model.film_data[name]
        ni = len(system.indices)
        torch.manual_seed(0)
        b = torch.randn(ni, dtype=fd.tdtype, device="cuda")
        x = kernels.chol_solve(system.chol, b.clone())
        S = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, system.indices_device,
                                    system.indices_device, sign=1.0, dtype=str(fd.dtype), row_scale=fd.w)
        worst = max(worst, float((kernels.gemv(S, ni, ni, x) - b).abs().max() / b.abs().max()))
    return worst


for case in wanted:
    if case == "mix":
        continue
    K, kinds, dz, dtype, method, iters = CASES[case]
    device = synthetic.make_stack_device(K, kinds, z_spacing=dz, solve_dtype=dtype)
    cc = {f"hole{i}": 1.5 for i, k in enumerate(kinds) if k == "washer"}
    ref, worst, bad, t0 = None, 0.0, 0, time.perf_counter()
    for rep in range(reps):
        model = sc.factorize_model(device=device, current_units="uA", method=method, circulating_currents=cc)
        if rep < 2:
            r = residuals(model)
            worst = max(worst, r)
            assert r < (1e-11 if dtype == "float64" else 2e-3), (case, rep, r)
        sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=iters)
        streams = [{nm: s.film_solutions[nm].stream.copy() for nm in device.films} for s in sols]
        if ref is None:
            ref = streams
        for it, (a, b) in enumerate(zip(streams, ref)):
            for nm in device.films:
                if not np.array_equal(a[nm], b[nm]):
                    bad += 1
                    d = np.abs(a[nm].astype(np.float64) - b[nm])
                    print(f"   {case} rep {rep}: iterate {it} film {nm} differs in {np.count_nonzero(d)} entries, "
                          f"max rel {d.max() / np.abs(b[nm]).max():.2e}", flush=True)
                    break
            else:
                continue
            break
        del model, sols
    failed += bad
    print(f"{case}: K={K} films={len(kinds)} {dtype} method={method}: {reps} cold factorizations + {iters}-iteration solves, "
          f"{'ALL bit-identical' if not bad else str(bad) + ' DIFFERENT'}; worst residual {worst:.1e}; "
          f"{1e3 * (time.perf_counter() - t0) / reps:.0f} ms per repetition", flush=True)

if "mix" in wanted:
    # routes and precisions interleaved in one process (the Cholesky and LU schedules share the chain-stream pool; every
    # change of route re-uses streams the other route just left)
    device64 = synthetic.make_stack_device(64, ("washer", "disk"), solve_dtype="float64")
    device32 = synthetic.make_stack_device(64, ("washer", "disk"), solve_dtype="float32")
    first, bad = {}, 0
    nmix = max(4, reps // 3)
    for rep in range(nmix):
        for dev, method in ((device64, "auto"), (device32, "auto"), (device64, "lu"), (device32, "lu")):
            model = sc.factorize_model(device=dev, current_units="uA", method=method)
            g = sc.solve(model=model, applied_field=sc.ConstantField(0.7), iterations=2)[-1].film_solutions["disk1"].stream
            key = (dev.solve_dtype, method)
            first.setdefault(key, g)
            if not (g == first[key]).all():
                bad += 1
                print(f"   mix rep {rep} {key}: differs", flush=True)
            del model
    failed += bad
    print(f"mix: {nmix} x (cholesky f64, cholesky f32, lu f64, lu f32) interleaved: "
          f"{'ALL bit-identical' if not bad else str(bad) + ' DIFFERENT'}", flush=True)
print(f"{failed} differing runs in all", flush=True)
sys.stdout.flush()
os._exit(1 if failed else 0)   # (not sys.exit: the disturber thread is still launching work)
