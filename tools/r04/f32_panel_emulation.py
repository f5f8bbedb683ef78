"""float32 factorization: which step of a 256-column blocked Cholesky sets the backward error?  The schedule of
chol.hip emulated with torch float32 products on the config-H disk film's matrix, one variant per candidate
(development aid; nothing here is product code):

  inv32      diagonal block factored and inverted in float32, panel L21 = A21 W^T        (what the kernels do)
  inv64      diagonal block factored and inverted in float64, both rounded to float32, panel L21 = A21 W^T
  trsm       diagonal block in float32, panel by substitution (LAPACK's way)
  inv32+fix  inv32, then one correction of the panel: L21 += (A21 - L21 L11^T) W^T
"""
import os, sys
import numpy as np
import scipy.linalg as la
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import superscreen_amd as sc
from superscreen_amd import kernels, synthetic

torch.backends.cuda.matmul.allow_tf32 = False
K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
NB = 256
dev32 = synthetic.make_stack_device(K, ("disk",), solve_dtype="float32")
m32 = sc.factorize_model(device=dev32, current_units="uA")
name = list(dev32.films)[0]
sysm, fd = m32.film_systems[name], m32.film_data[name]
ni = len(sysm.indices)
ix = sysm.indices_device
S32 = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix, ix, sign=1.0, dtype="float32", row_scale=fd.w)[:ni, :ni].contiguous()
S64 = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix, ix, sign=1.0, dtype="float64", row_scale=fd.w)[:ni, :ni].contiguous()
S32d = S32.double()
b64 = torch.ones(ni, dtype=torch.float64, device="cuda")
x_ref = torch.linalg.solve(S64, b64)
del S64


def blocked(variant):
    A = S32.clone()
    n = A.shape[0]
    eye = torch.eye(NB, device="cuda")
    for c in range(0, n, NB):
        e = min(c + NB, n)
        D = A[c:e, c:e]
        if variant == "inv64":
            L11d = torch.linalg.cholesky(D.double())
            Wd = torch.linalg.solve_triangular(L11d, torch.eye(e - c, device="cuda", dtype=torch.float64), upper=False)
            L11, W = L11d.float(), Wd.float()
        else:
            L11 = torch.linalg.cholesky(D)
            W = torch.linalg.solve_triangular(L11, eye[: e - c, : e - c], upper=False)
        A[c:e, c:e] = L11
        if e == n:
            break
        A21 = A[e:, c:e]
        if variant == "trsm":
            P = torch.linalg.solve_triangular(L11, A21.T, upper=False).T.contiguous()
        else:
            P = A21 @ W.T
            if variant == "inv32+fix":
                P = P + (A21 - P @ L11.T) @ W.T
        A[e:, c:e] = P
        A[e:, e:] -= P @ P.T
    return torch.tril(A)


def report(tag, L):
    Ld = L.double()
    back = float((S32d - Ld @ Ld.T).abs().max() / S32d.abs().max())
    y = torch.linalg.solve_triangular(Ld, b64[:, None], upper=False)
    x = torch.linalg.solve_triangular(Ld.T, y, upper=True)[:, 0]
    err = float((x - x_ref).abs().max() / x_ref.abs().max())
    print(f"{tag:>12}: backward error {back:.2e}   solve error (float64 substitution) {err:.2e}", flush=True)


print(f"n_i = {ni}")
report("GPU kernels", torch.tril(sysm.chol.L[:ni, :ni]))
report("spotrf", torch.from_numpy(la.cholesky(S32.cpu().numpy(), lower=True)).cuda())
for v in ("inv32", "inv64", "trsm", "inv32+fix"):
    report(v, blocked(v))
