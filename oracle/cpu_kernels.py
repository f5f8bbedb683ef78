"""ctypes access to ``oracle/_build/liboracle_kernels.so`` (TEST INFRASTRUCTURE).

OpenMP C versions of the reference's two numba ``prange`` kernels; used to cross-check the
numpy oracle and as the multi-threaded CPU baseline in ``bench.py``."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle_kernels.so")
_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_PATH)
        _lib.oracle_q_matrix.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _lib.oracle_biot_savart.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                            ctypes.c_double, ctypes.c_int64, ctypes.c_void_p]
    return _lib


def q_matrix(points: np.ndarray) -> np.ndarray:
    """distance.py:87-115."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    n = len(points)
    out = np.empty((n, n), dtype=np.float64)
    _load().oracle_q_matrix(points.ctypes.data, n, out.ctypes.data)
    return out


def biot_savart_film_to_film(*, film1_sites, film1_z0, film1_areas, film1_J, film2_sites, film2_z0):
    """solver/solve.py:28-73."""
    s1 = np.ascontiguousarray(film1_sites, dtype=np.float64)
    a1 = np.ascontiguousarray(film1_areas, dtype=np.float64)
    J1 = np.ascontiguousarray(film1_J, dtype=np.float64)
    s2 = np.ascontiguousarray(film2_sites, dtype=np.float64)
    out = np.empty(len(s2), dtype=np.float64)
    _load().oracle_biot_savart(s1.ctypes.data, float(film1_z0), a1.ctypes.data, J1.ctypes.data,
                               len(s1), s2.ctypes.data, float(film2_z0), len(s2), out.ctypes.data)
    return out
