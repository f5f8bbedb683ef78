"""Calibration workload for the FETCH_SIZE / WRITE_SIZE counters on the SYRK kernel's access
patterns (run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`, then `--pmc WRITE_SIZE`).

Launch 1: lower-tile SYRK, M = 16384, K = 16   -> traffic ~ C tiles read + written, panel negligible
Launch 2: lower-tile SYRK, M = 16384, K = 256  -> + panel reads (LDS-DMA, 16 B / lane)
Launch 3: row-major GEMV 16384 x 16384 (16 B / lane streaming read of a known 2.147 GB matrix)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superscreen_amd import kernels  # noqa: E402

M = 16384
dev = torch.device("cuda", 0)
C = torch.zeros(M, M, dtype=torch.float64, device=dev)
P = torch.randn(M, 256, dtype=torch.float64, device=dev) * 1e-3
x = torch.ones(M, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for K in (16, 256):
    kernels.gemm_ex(0, 1, True, P, P, C, M, M, K, alpha=-1.0, beta=1.0)
    torch.cuda.synchronize()
y = kernels.gemv(C, M, M, x)
torch.cuda.synchronize()
tiles = (M // 128) * (M // 128 + 1) // 2
print(f"C lower tiles: {tiles} x 128 KiB = {tiles * 131072 / 1e6:.1f} MB read and written per SYRK; "
      f"panel {M * 256 * 8 / 1e6:.1f} MB; GEMV matrix {M * M * 8 / 1e6:.1f} MB")
