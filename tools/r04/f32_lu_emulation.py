"""float32 LU route: which step of the 256-column blocked LU without interchanges sets the error?  The schedule of
lu.hip emulated with torch float32 products on the config-H disk film's matrix -A (development aid):

  inv32   diagonal block factored (no pivoting) and both factors inverted in float32, L21 = A21 U11^-1, U12 = L11^-1 A12
  inv64   the diagonal block's factorization and inverses in float64, rounded to float32
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
m32 = sc.factorize_model(device=dev32, current_units="uA", method="lu")
name = list(dev32.films)[0]
sysm, fd = m32.film_systems[name], m32.film_data[name]
ni = len(sysm.indices)
ix = sysm.indices_device
A32 = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix, ix, sign=-1.0, dtype="float32")[:ni, :ni].contiguous()
A64 = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix, ix, sign=-1.0, dtype="float64")[:ni, :ni].contiguous()
b64 = torch.ones(ni, dtype=torch.float64, device="cuda")
x_ref = torch.linalg.solve(A64, b64)
A32d = A32.double()
del A64


def blocked(variant):
    A = A32.clone()
    n = A.shape[0]
    for c in range(0, n, NB):
        e = min(c + NB, n)
        D = A[c:e, c:e]
        dt = torch.float64 if variant == "inv64" else torch.float32
        LU, _ = torch.linalg.lu_factor(D.to(dt), pivot=False)
        eye = torch.eye(e - c, device="cuda", dtype=dt)
        L11 = torch.tril(LU, -1) + eye
        U11 = torch.triu(LU)
        WL = torch.linalg.solve_triangular(L11, eye, upper=False, unitriangular=True).float()
        WU = torch.linalg.solve_triangular(U11, eye, upper=True).float()
        A[c:e, c:e] = LU.float()
        if e == n:
            break
        L21 = A[e:, c:e] @ WU
        U12 = WL @ A[c:e, e:]
        A[e:, c:e] = L21
        A[c:e, e:] = U12
        A[e:, e:] -= L21 @ U12
    return A


def report(tag, LU):
    LUd = LU.double()
    L = torch.tril(LUd, -1) + torch.eye(ni, device="cuda", dtype=torch.float64)
    U = torch.triu(LUd)
    back = float((A32d - L @ U).abs().max() / A32d.abs().max())
    y = torch.linalg.solve_triangular(L, b64[:, None], upper=False, unitriangular=True)
    x = torch.linalg.solve_triangular(U, y, upper=True)[:, 0]
    err = float((x - x_ref).abs().max() / x_ref.abs().max())
    print(f"{tag:>12}: backward error {back:.2e}   solve error (float64 substitution) {err:.2e}", flush=True)


print(f"n_i = {ni}")
lu_gpu = sysm.factors
report("GPU kernels", lu_gpu.lu[:ni, :ni])
print("GPU route interchanges:", int((lu_gpu.ipiv[:ni].cpu().numpy() != np.arange(ni)).sum()))
lu_h, piv = la.lu_factor(A32.cpu().numpy())
print("sgetrf interchanges:", int((piv != np.arange(ni)).sum()))
report("sgetrf", torch.from_numpy(lu_h).cuda())
for v in ("inv32", "inv64"):
    report(v, blocked(v))
x_gpu = kernels.lu_solve(lu_gpu, b64.float().clone()).double()[:ni]
if True:
    print(f"GPU float32 LU route, its own solve: {float((x_gpu - x_ref).abs().max() / x_ref.abs().max()):.2e}")
x_lap = torch.from_numpy(la.lu_solve((lu_h, piv), np.ones(ni, dtype=np.float32)).astype(np.float64)).cuda()
print(f"sgetrf + sgetrs: {float((x_lap - x_ref).abs().max() / x_ref.abs().max()):.2e}")
