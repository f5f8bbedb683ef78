"""SYRK (lower tiles, C -= P P^T) time vs trailing order M at K = 256 / 512: tile quantisation and per-launch
overhead of the trailing update as the factorization sees it (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from superscreen_amd import kernels as K
Ms = [int(a) for a in sys.argv[1:]] or [1024, 2048, 3072, 4096, 5120, 6144, 8192, 9216, 10240, 12288, 14336, 16384, 18176, 20224]
Cbuf = torch.randn((max(Ms), max(Ms)), dtype=torch.float64, device="cuda")
for M in Ms:
    for Kd in (256, 512):
        P = torch.randn((M, Kd), dtype=torch.float64, device="cuda")
        C = Cbuf[:M]
        for _ in range(2):
            K.gemm_ex(0, 1, True, P, P, C, M, M, Kd, alpha=-1.0, beta=1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 8
        e0.record()
        for _ in range(reps):
            K.gemm_ex(0, 1, True, P, P, C, M, M, Kd, alpha=-1e-3, beta=1.0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nt = M // 128
        tiles = nt * (nt + 1) // 2
        print(f"M={M:6d} K={Kd:4d} tiles={tiles:6d} rounds512={tiles / 512:6.2f}: {ms * 1e3:8.1f} us -> "
              f"{Kd * M * (M + 128) / ms / 1e9:6.1f} TFLOP/s (tiles), us/tile-round {ms * 1e3 / max(1.0, tiles / 512):7.1f}", flush=True)
        del P
