"""SYRK rate vs K, beta and M (how much of the trailing update is C-tile traffic / tile quantisation?)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from superscreen_amd import kernels as K
for M in (16384, 9216, 4096):
    for Kd in (256, 512, 1024, 2048):
        if M != 16384 and Kd > 512:
            continue
        P = torch.randn((M, Kd), dtype=torch.float64, device="cuda")
        C = torch.randn((M, M), dtype=torch.float64, device="cuda")
        for beta in (1.0, 0.0):
            for _ in range(2):
                K.gemm_ex(0, 1, True, P, P, C, M, M, Kd, alpha=-1.0, beta=beta)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                K.gemm_ex(0, 1, True, P, P, C, M, M, Kd, alpha=-1.0 if beta else 1e-3, beta=beta)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"M={M:6d} K={Kd:5d} beta={beta}: {ms:7.3f} ms -> {Kd * M * (M + 128) / ms / 1e9:6.1f} TFLOP/s (tiles computed), "
                  f"{Kd * M * (M + 1) / ms / 1e9:6.1f} algorithmic", flush=True)
        del P, C
