"""Cold factorizations of the films of a stack, over and over, each result compared bit for bit with the first one; a
difference is located (which film, which part of the factor buffer / aux buffer, first rows and columns).

    python tools/chol_race_hunt.py [reps=100] [K=100] [films=4]
    SSA_CHOL_DEBUG=split=0,tail=0,late=1,delay=0,sync=1   schedule variants (chol.hip: CholDebug)

The systems are assembled once per repetition exactly as factorize_linear_systems does (lower tiles only, padded
buffer from torch.empty), factored by kernels.chol_factor_batch; no solve.
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
from superscreen_amd.solver import FilmDeviceData, make_film_info  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nfilms = int(sys.argv[3]) if len(sys.argv) > 3 else 4
kinds = ("disk",) * nfilms
device = synthetic.make_stack_device(K, kinds, z_spacing=0.5, solve_dtype="float64")
names = list(device.films)
dtype = device.solve_dtype
info = make_film_info(device=device, vortices=[], circulating_currents={}, terminal_currents={})
fds = {nm: FilmDeviceData(info[nm], device.meshes[nm], dtype, False) for nm in names}
ix = {nm: torch.from_numpy(info[nm].interior_indices.astype(np.int64)).cuda() for nm in names}
ni = {nm: len(info[nm].interior_indices) for nm in names}


def factor_all():
    systems = []
    for nm in names:
        fd = fds[nm]
        npad = kernels.chol_padded_n(ni[nm])
        S = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix[nm], ix[nm], sign=1.0, dtype=dtype,
                                    row_scale=fd.w, lower_only=True, ld=kernels.padded_ld(npad, dtype), alloc_rows=npad)
        systems.append((S, ni[nm]))
    return kernels.chol_factor_batch(systems)


def locate(a, b, what):
    d = a != b
    if a.dim() == 1:
        idx = d.nonzero().flatten()
        return f"{what}: {idx.numel()} words differ, first {int(idx[0])}, last {int(idx[-1])}"
    rows = d.any(dim=1).nonzero().flatten()
    cols = d.any(dim=0).nonzero().flatten()
    lower = torch.tril(d).sum().item()
    upper = torch.triu(d, 1).sum().item()
    diff = (a[d] - b[d]).abs()
    rel = float((diff / b[d].abs().clamp_min(1e-300)).max())
    return (f"{what}: {int(d.sum())} entries differ (lower {lower}, upper {upper}); rows {int(rows[0])}..{int(rows[-1])} "
            f"({rows.numel()}), cols {int(cols[0])}..{int(cols[-1])} ({cols.numel()}); first rows {rows[:6].tolist()}, "
            f"first cols {cols[:6].tolist()}; max |diff| {float(diff.max()):.3e}, max rel {rel:.3e}")


print(f"SSA_CHOL_DEBUG={os.environ.get('SSA_CHOL_DEBUG', '')!r} poison={poison.mode() or 'off'} K={K} films={nfilms} "
      f"unknowns={[ni[nm] for nm in names]} reps={reps}", flush=True)
tracing = "trace=1" in os.environ.get("SSA_CHOL_DEBUG", "")
TRACE_FILE = "/tmp/chol_trace.bin"
if tracing:
    os.environ["SSA_CHOL_TRACE_FILE"] = TRACE_FILE


def read_trace():
    """[film][panel][tile (I, K), K <= I, 136 of them][256]: the diagonal block of every ROUND as its workgroup read it."""
    t = np.fromfile(TRACE_FILE, dtype=np.float64)
    return t.reshape(nfilms, -1, 136, 256)


def trace_report(t, t0):
    lines = []
    for i in range(nfilms):
        for p in range(t.shape[1]):
            d = (t[i, p] != t0[i, p]).any(axis=1)
            if d.any():
                tiles = []
                for tid in np.flatnonzero(d):
                    I = int((np.sqrt(8 * tid + 1) - 1) // 2)
                    tiles.append((I, int(tid - I * (I + 1) // 2)))
                diff = np.abs(t[i, p] - t0[i, p]).max()
                lines.append(f"   trace: film {i} panel {p} (column {256 * p}): INPUT block differs in {len(tiles)} tiles, "
                             f"first (row, col) tiles {tiles[:8]}, max |diff| {diff:.3e}")
    return lines or ["   trace: every diagonal block was READ with the reference's bits"]


ref = factor_all()
torch.cuda.synchronize()
assert all(f.info == 0 for f in ref)
refs = [(f.L.clone(), f.aux.clone()) for f in ref]
trace0 = read_trace() if tracing else None
del ref
bad = 0
t0 = time.perf_counter()
for rep in range(reps):
    fs = factor_all()
    torch.cuda.synchronize()
    for i, (f, (L0, aux0)) in enumerate(zip(fs, refs)):
        n = f.n
        same_L = torch.equal(f.L[:n, :n], L0[:n, :n])
        # aux = inverse blocks | their transposes | scratch (chol.hip: aux_layout); the scratch part may hold words no
        # kernel writes, so only the 2 x nblk x 4096^2 words the solves read are compared
        npad = kernels.chol_padded_n(n)
        used = 2 * ((npad + 4095) // 4096) * 4096 * 4096
        same_aux = torch.equal(f.aux[:used], aux0[:used])
        if not (same_L and same_aux):
            bad += 1
            print(f"rep {rep} film {i} ({names[i]}): DIFFERENT  info={f.info}", flush=True)
            if not same_L:
                print("   " + locate(f.L[:n, :n], L0[:n, :n], "factor"), flush=True)
            if not same_aux:
                print("   " + locate(f.aux[:used], aux0[:used], "aux (inverse blocks)"), flush=True)
            if tracing:
                print("\n".join(trace_report(read_trace(), trace0)[:12]), flush=True)
    del fs
print(f"{bad} differing factorizations in {reps} x {nfilms}; {1e3 * (time.perf_counter() - t0) / reps:.0f} ms per repetition",
      flush=True)
sys.exit(1 if bad else 0)
