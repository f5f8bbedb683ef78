"""float32: where does the remaining error sit -- factorization (backward error of L L^T against S) or the solves
(explicit block inverses)?  Config-H disk film (20 419 unknowns), GPU route against LAPACK spotrf / spotrs on the host
(development aid)."""
import os, sys, time
import numpy as np
import scipy.linalg as la
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import superscreen_amd as sc
from superscreen_amd import kernels, synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 91
dev32 = synthetic.make_stack_device(K, ("disk",), solve_dtype="float32")
m32 = sc.factorize_model(device=dev32, current_units="uA")
name = list(dev32.films)[0]
sysm, fd = m32.film_systems[name], m32.film_data[name]
ni = len(sysm.indices)
ix = sysm.indices_device
# the matrix the Cholesky route factors, in float32 as assembled, and in float64
S32 = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix, ix, sign=1.0, dtype="float32", row_scale=fd.w)[:ni, :ni]
S64 = kernels.system_assemble(fd.xy, fd.w, fd.qdiag, fd.Lambda, *fd.lap, ix, ix, sign=1.0, dtype="float64", row_scale=fd.w)[:ni, :ni]
S32d = S32.double()
print(f"n_i = {ni}; |S32 - S64| / |S64| = {float((S32d - S64).abs().max() / S64.abs().max()):.2e}")
f = sysm.chol
L = torch.tril(f.L[:ni, :ni]).double()
R = S32d - L @ L.T
print(f"GPU float32 factor: max|S32 - L L^T| / max|S32| = {float(R.abs().max() / S32d.abs().max()):.2e}")
Lh = la.cholesky(S32.cpu().numpy(), lower=True)              # LAPACK spotrf
Lhd = torch.from_numpy(Lh.astype(np.float64)).cuda()
Rh = S32d - Lhd @ Lhd.T
print(f"LAPACK spotrf     : max|S32 - L L^T| / max|S32| = {float(Rh.abs().max() / S32d.abs().max()):.2e}")
# solves with a smooth right-hand side: x_true from the float64 matrix
torch.manual_seed(0)
b64 = torch.ones(ni, dtype=torch.float64, device="cuda")
x_ref = torch.linalg.solve(S64, b64)
b32 = b64.float()
x_gpu = kernels.chol_solve(f, b32.clone()).double()
x_lap = torch.from_numpy(la.cho_solve((Lh, True), b32.cpu().numpy()).astype(np.float64)).cuda()
# GPU factor, exact triangular solves in float64 (what the explicit block inverses of the solve phase cost)
y = torch.linalg.solve_triangular(L, b64[:, None], upper=False)
x_mix = torch.linalg.solve_triangular(L.T, y, upper=True)[:, 0]
rel = lambda a: float((a - x_ref).abs().max() / x_ref.abs().max())
print(f"solve error against the float64 solution: GPU float32 route {rel(x_gpu):.2e} | LAPACK spotrf + spotrs {rel(x_lap):.2e} | "
      f"GPU float32 FACTOR with float64 substitution {rel(x_mix):.2e} | float32 rounding of S alone (float64 solve of S32) "
      f"{rel(torch.linalg.solve(S32d, b64)):.2e}")
