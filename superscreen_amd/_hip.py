"""ctypes binding of ``libsuperscreen_hip.so`` (the C ABI in ``include/superscreen_hip.h``).

There is no CPU fallback: if the library is missing or a call fails, a
:class:`HipLibraryError` is raised.  PyTorch is used only as plumbing (device memory via
``torch.empty(..., device="cuda")`` and the current HIP stream handle).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p
from typing import Optional

_PKG = os.path.dirname(os.path.abspath(__file__))
# SSA_LIB_PATH: an experiment build of the same C ABI (python -m superscreen_amd.build -D... --libname=...)
LIB_PATH = os.environ.get("SSA_LIB_PATH") or os.path.join(_PKG, "lib", "libsuperscreen_hip.so")

SSA_F32 = 0
SSA_F64 = 1
ABI_VERSION = 6   # SSA_ABI_VERSION of include/superscreen_hip.h


class HipLibraryError(RuntimeError):
    """The HIP library is missing, could not be loaded, or a kernel call failed."""


P = c_void_p
I64 = c_int64

# name -> (restype, argtypes); mirrors include/superscreen_hip.h one to one.
SIGNATURES = {
    "ssa_abi_version": (c_int, []),
    "ssa_error_string": (c_char_p, [c_int]),
    "ssa_device_info": (c_int, [P, P, P, c_int]),
    "ssa_q_assemble": (c_int, [P, P, P, I64, P, I64, c_int, P, P]),
    "ssa_system_assemble_workspace_bytes": (c_size_t, [I64, I64, I64]),
    "ssa_system_assemble": (c_int, [P, P, P, P, I64, P, P, P, P, I64, P, I64, c_double, P, c_int,
                                    P, I64, c_int, P, c_size_t, P]),
    "ssa_chol_padded_n": (c_int64, [I64]),
    "ssa_chol_aux_bytes": (c_size_t, [I64, c_int]),
    "ssa_chol_factor": (c_int, [P, I64, I64, P, P, c_int, P]),
    "ssa_chol_factor_batch": (c_int, [c_int, P, P, P, P, P, c_int, P]),
    "ssa_chol_solve_workspace_bytes": (c_size_t, [I64, I64, c_int]),
    "ssa_chol_solve": (c_int, [P, I64, I64, P, P, I64, I64, c_int, P, c_size_t, P]),
    "ssa_chol_chain_stream_costs": (c_int, [P, P, c_int]),
    "ssa_chol_solve_batch": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P, P, P]),
    "ssa_chol_factor_batch_blk": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P]),
    "ssa_chol_solve_blk": (c_int, [P, I64, I64, P, P, I64, I64, c_int, P, c_size_t, c_int, P]),
    "ssa_chol_solve_batch_blk": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P, P, c_int, P]),
    "ssa_gemm_ex": (c_int, [c_int, c_int, c_int, I64, I64, I64, c_double, P, I64, P, I64, c_double,
                            P, I64, c_int, P]),
    "ssa_profile_read": (c_int, [c_int, P, P, P]),
    "ssa_lu_factor_workspace_bytes": (c_size_t, [I64, c_int]),
    "ssa_lu_aux_bytes": (c_size_t, [I64, c_int]),
    "ssa_lu_factor": (c_int, [P, I64, I64, P, P, P, c_int, P, c_size_t, P]),
    "ssa_lu_pivots_to_permutation": (c_int, [P, I64, P]),
    "ssa_lu_padded_n": (c_int64, [I64]),
    "ssa_lu_factor_nopivot_workspace_bytes": (c_size_t, [I64, c_int]),
    "ssa_lu_factor_nopivot_batch": (c_int, [c_int, P, P, P, P, P, P, c_int, P, P, P]),
    "ssa_lu_solve_workspace_bytes": (c_size_t, [I64, I64, c_int]),
    "ssa_lu_solve": (c_int, [P, I64, I64, P, P, I64, I64, c_int, P, c_size_t, P]),
    "ssa_gemv": (c_int, [P, I64, I64, I64, P, P, P, P, c_double, c_double, c_int, P]),
    "ssa_row_scale": (c_int, [P, P, P, I64, I64, c_int, P]),
    "ssa_self_field_workspace_bytes": (c_size_t, [I64]),
    "ssa_self_field": (c_int, [P, P, P, P, I64, P, c_double, c_int, P, c_size_t, P]),
    "ssa_self_field_rows": (c_int, [P, P, P, P, I64, P, I64, P, c_double, c_int, P, c_size_t, P]),
    "ssa_london_field_rows": (c_int, [P, P, P, P, P, P, P, P, I64, I64, P, c_int, P]),
    "ssa_film_rhs": (c_int, [P, P, P, P, I64, I64, P, c_int, P]),
    "ssa_scatter_add": (c_int, [P, P, P, I64, I64, c_int, P]),
    "ssa_index_add_scalar": (c_int, [P, P, I64, P, I64, c_int, P]),
    "ssa_current_density": (c_int, [P, P, P, P, P, I64, I64, P, c_int, P]),
    "ssa_scale": (c_int, [P, P, c_double, I64, c_int, P]),
    "ssa_biot_savart_workspace_bytes": (c_size_t, [I64]),
    "ssa_biot_savart": (c_int, [P, P, P, I64, I64, I64, P, I64, c_double, P, c_int, c_int, P,
                                c_size_t, P]),
    "ssa_gemm": (c_int, [I64, I64, I64, c_double, P, I64, P, I64, c_double, P, I64, c_int, P]),
    "ssa_sheet_field_workspace_bytes": (c_size_t, [I64, c_int]),
    "ssa_sheet_field": (c_int, [P, P, P, I64, c_double, P, I64, c_double, c_int, P, P, c_size_t, P]),
    "ssa_sheet_potential": (c_int, [P, P, P, I64, c_double, P, I64, c_double, P, P, c_size_t, P]),
    "ssa_pairwise_multi_workspace_bytes": (c_size_t, [I64]),
    "ssa_self_field_multi": (c_int, [P, P, P, P, I64, I64, P, c_double, c_int, P, c_size_t, P]),
    "ssa_biot_savart_multi_rows": (c_int, [P, P, P, I64, P, I64, P, I64, c_double, I64, P, c_int, c_int, P, c_size_t, P]),
    "ssa_self_field_multi_rows": (c_int, [P, P, P, P, I64, I64, P, I64, P, c_double, c_int, P, c_size_t, P]),
    "ssa_biot_savart_multi": (c_int, [P, P, P, I64, P, I64, c_double, I64, P, c_int, c_int, P, c_size_t, P]),
    "ssa_fill_probe": (c_int, [P, c_size_t, P]),
    "ssa_mfma_probe": (c_int, [c_int, P, P, P]),
    "ssa_chol_chain_streams_invalidate": (c_int, []),
    "ssa_profile_begin": (c_int, []),
    "ssa_profile_begin_kinds": (c_int, [ctypes.c_uint]),
    "ssa_profile_end": (c_int, []),
    "ssa_rccl_unique_id": (c_int, [P]),
    "ssa_rccl_comm_create": (c_int, [P, c_int, c_int, P]),
    "ssa_rccl_comm_destroy": (c_int, [P]),
    "ssa_coupling_allreduce": (c_int, [P, I64, c_int, P, P]),
    "ssa_shutdown": (c_int, []),
}

_lib: Optional[ctypes.CDLL] = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Loads the shared library and checks that every declared symbol is exported."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} not found. Build it with `python -m superscreen_amd.build` "
            "(needs hipcc, targets gfx950). There is no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # missing ROCm runtime etc.
        raise HipLibraryError(f"Could not load {path}: {e}") from e
    # the version first: a stale build (or another one selected through SSA_LIB_PATH) then fails with a
    # version message instead of a missing-symbol error further down
    try:
        lib.ssa_abi_version.restype = c_int
        lib.ssa_abi_version.argtypes = []
        version = lib.ssa_abi_version()
    except AttributeError as e:
        raise HipLibraryError(f"{path} does not export ssa_abi_version: not a superscreen_amd library") from e
    if version != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: {path} reports {version}, this package binds version "
                              f"{ABI_VERSION} (include/superscreen_hip.h). Rebuild with `python -m superscreen_amd.build`.")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load_library().ssa_error_string(status).decode()
        raise HipLibraryError(f"{what} failed with status {status}: {msg}")


def dtype_code(dtype) -> int:
    import numpy as np

    dt = np.dtype(str(dtype).replace("torch.", ""))
    if dt == np.float64:
        return SSA_F64
    if dt == np.float32:
        return SSA_F32
    raise ValueError(f"Unsupported solve dtype {dtype!r} (float32 or float64).")


def ptr(t) -> Optional[int]:
    """Device pointer of a CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("Expected a device tensor.")
    if not t.is_contiguous():
        raise HipLibraryError("Expected a contiguous tensor.")
    return t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


def require_gpu() -> None:
    import torch

    if not torch.cuda.is_available():
        raise HipLibraryError(
            "No HIP device is visible (torch.cuda.is_available() is False). "
            "superscreen_amd has no CPU fallback for the solver hot path."
        )
