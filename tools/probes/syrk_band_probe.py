"""SYRK (lower tiles) rate by band height of the tile order (SSA_SYRK_BAND, read once per process: run one process per
value), with the panel compact ([M, K] buffer) and as it lies in a factorization (K columns of an M_total-wide
matrix)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from superscreen_amd import kernels as K
big = torch.randn((20480, 20480), dtype=torch.float64, device="cuda")
out = []
for M in (12288, 16384, 20224):
    for Kd in (256, 512):
        for layout in ("compact", "in place"):
            if layout == "compact":
                P, ldp = torch.randn((M, Kd), dtype=torch.float64, device="cuda"), Kd
                Cm = big[:M]
                args = lambda: K.gemm_ex(0, 1, True, P, P, Cm, M, M, Kd, alpha=-1e-3, beta=1.0)
            else:
                # panel = columns [0, Kd) of rows [20480 - M, 20480), C = the block behind it (as in chol.hip)
                from superscreen_amd import _hip
                lib = _hip.load_library()
                r0 = 20480 - M
                base = big.data_ptr()
                pa = base + (r0 * 20480) * 8
                pc = base + (r0 * 20480 + Kd) * 8
                Mm = M - Kd
                args = lambda: _hip.check(lib.ssa_gemm_ex(0, 1, 1, Mm, Mm, Kd, -1e-3, pa + Kd * 20480 * 8, 20480, pa + Kd * 20480 * 8, 20480, 1.0,
                                                          pc + Kd * 20480 * 8, 20480, 1, _hip.current_stream()), "gemm")
            for _ in range(2):
                args()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                args()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            Me = M if layout == "compact" else M - Kd
            out.append(f"M={Me:6d} K={Kd} {layout:9s}: {ms * 1e3:8.1f} us {Kd * Me * (Me + 128) / ms / 1e9:6.1f} TF")
print(f"band={os.environ.get('SSA_SYRK_BAND', '8')}: " + " | ".join(out))
