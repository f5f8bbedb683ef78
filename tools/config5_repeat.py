"""Config 5 (4-film stack, 30 301 vertices per film) over and over: cold factorization + solve in the two self-field
modes, every iterate of every film compared BIT FOR BIT with the first run, the factor buffers compared through a
checksum.  The tail of tests/test_headline_gpu.py::test_full_size_london_system[config5...] in a loop, with the place
of the first difference reported (iterate 0 = factorization / solve kernels, later = coupling path).

    python tools/config5_repeat.py [reps=10] [K=100] [films=4] [iters=2]
    SSA_POISON=nan|big ...   uninitialised device buffers filled (tools/poison.py)
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
from superscreen_amd import synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nfilms = int(sys.argv[3]) if len(sys.argv) > 3 else 4
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
kinds = ("disk",) * nfilms
device = synthetic.make_stack_device(K, kinds, z_spacing=0.5, solve_dtype="float64")
names = list(device.films)


def checksum(t):
    """Order-independent exact checksum of a tensor's bits (wrapping int64 sum of the words)."""
    v = t.contiguous().view(torch.int64) if t.element_size() == 8 else t.contiguous().view(torch.int32).to(torch.int64)
    return int(v.sum().item())


def factor_sums(model):
    out = {}
    for nm in names:
        ch = model.film_systems[nm].chol
        n = ch.n
        L = ch.L[:n, :n]
        out[nm] = (checksum(torch.tril(L)), checksum(torch.triu(L, 1)), checksum(ch.aux[: ch.aux.numel()]))
    return out


def run(mode, sums=True):
    model = sc.factorize_model(device=device, current_units="uA", self_field=mode)
    sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=iters)
    streams = [{nm: s.film_solutions[nm].stream.copy() for nm in names} for s in sols]
    others = [{nm: (None if s.film_solutions[nm].field_from_other_films is None
                    else s.film_solutions[nm].field_from_other_films.copy()) for nm in names} for s in sols]
    fs = factor_sums(model) if sums else None
    return model, streams, others, fs


def first_difference(streams, ref):
    for it, (a, b) in enumerate(zip(streams, ref)):
        for nm in names:
            if not np.array_equal(a[nm], b[nm]):
                d = np.abs(a[nm] - b[nm])
                return it, nm, int(np.count_nonzero(d)), float(d.max() / np.abs(b[nm]).max()), int(np.argmax(d))
    return None


print(f"poison={poison.mode() or 'off'} K={K} films={nfilms} iters={iters} reps={reps} "
      f"device={torch.cuda.get_device_name(0)} CUs={torch.cuda.get_device_properties(0).multi_processor_count}", flush=True)
model, ref_streams, ref_others, ref_sums = run("auto")
finite = all(np.isfinite(s[nm]).all() for s in ref_streams for nm in names)
print(f"reference run (auto): finite={finite} max|g|={max(np.abs(ref_streams[-1][nm]).max() for nm in names):.6e}", flush=True)
# the same model solved again (warm): must be bit-identical
again = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=iters)
w = first_difference([{nm: s.film_solutions[nm].stream for nm in names} for s in again], ref_streams)
print(f"warm re-solve of the reference model: {'identical' if w is None else w}", flush=True)
del model, again
torch.cuda.empty_cache()

bad = 0
for rep in range(reps):
    for mode in ("auto", "matrix_free"):
        t0 = time.perf_counter()
        model, streams, others, sums = run(mode)
        dt = time.perf_counter() - t0
        diff = first_difference(streams, ref_streams)
        fdiff = [nm for nm in names if sums[nm] != ref_sums[nm]]
        which = [tuple(i for i in range(3) if sums[nm][i] != ref_sums[nm][i]) for nm in fdiff]
        odiff = None
        for it in range(1, len(others)):
            for nm in names:
                if not np.array_equal(others[it][nm], ref_others[it][nm]):
                    odiff = (it, nm)
                    break
            if odiff:
                break
        ok = diff is None and not fdiff
        bad += 0 if ok else 1
        print(f"rep {rep} {mode:12s} {dt * 1e3:7.0f} ms  stream: {'identical' if diff is None else diff}  "
              f"factor: {'identical' if not fdiff else list(zip(fdiff, which))}  "
              f"coupling: {'identical' if odiff is None else odiff}", flush=True)
        if not ok and diff is not None:
            # is the difference reproducible on this very model (solve path) or a property of its factorization?
            again = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=iters)
            w = first_difference([{nm: s.film_solutions[nm].stream for nm in names} for s in again], streams)
            print(f"      re-solve of the differing model vs its own first solve: {'identical' if w is None else w}", flush=True)
        del model
        torch.cuda.empty_cache()
print(f"{bad} of {2 * reps} runs differ from the reference run", flush=True)
sys.exit(1 if bad else 0)
