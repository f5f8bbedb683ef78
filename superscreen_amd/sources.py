"""Applied-field sources.  The uniform field is the one on the BASELINE path
(``sources/constant.py:7-32``); the field of a current sheet (``sources/current.py``, SURVEY.md
section 8f row 3) runs on the GPU through ``ssa_sheet_field``.  Dipoles and Pearl vortices are out
of scope."""
from __future__ import annotations

from typing import Optional

import numpy as np

from .parameter import Parameter
from .units import parse_units


def constant(x, y, z, value=0):
    """Constant field (``sources/constant.py:7-20``)."""
    return value * np.ones_like(x, dtype=float)


def ConstantField(value: float = 0) -> Parameter:
    """A Parameter returning ``value`` at all ``x, y, z`` (``sources/constant.py:23-32``)."""
    return Parameter(constant, value=float(value))


def biot_savart_2d(x, y, z, *, positions: np.ndarray, current_densities: np.ndarray, z0: float = 0,
                   areas: Optional[np.ndarray] = None, length_units: str = "um", current_units: str = "uA",
                   vector: bool = True) -> np.ndarray:
    """Magnetic field in tesla of a sheet of current at height ``z0`` evaluated at ``(x, y, z)``
    (``sources/current.py:113-199``): ``positions`` ``(m, 2)`` and ``areas`` in ``length_units``,
    ``current_densities`` ``(m, 2)`` in ``current_units / length_units``.  Returns ``(n, 3)`` if
    ``vector`` else the z component ``(n,)``.  The all-pairs sum runs on the GPU
    (``ssa_sheet_field``, replacing the numba kernels at ``:13-110``)."""
    import torch

    from . import _hip, kernels

    _hip.require_gpu()
    to_meter = parse_units(length_units).scale
    to_amp_per_meter = parse_units(f"{current_units} / {length_units}").scale
    x, y, z = np.atleast_1d(x, y, z)
    if z.shape[0] == 1:
        z = z * np.ones_like(x)
    eval_xyz = np.ascontiguousarray(np.array([x, y, z], dtype=np.float64).T)
    positions, current_densities = np.atleast_2d(positions, current_densities)
    if areas is None:
        # triangulate the sheet to give every vertex an effective area (:184-187)
        from scipy.spatial import Delaunay

        from .fem import vertex_areas

        areas = vertex_areas(positions, Delaunay(positions).simplices)
    dev = torch.device("cuda", torch.cuda.current_device())

    def put(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)

    # B [T] = mu_0 / (4 pi) * sum a_k [m^2] J_k [A/m] d [m] / r^3 [m^3]
    #       = mu_0 / (4 pi) * (A/m per current/length unit) * (the same sum in the caller's units).
    # Like the reference module (sources/current.py:5) this takes mu_0 from scipy.constants, which
    # is CODATA 2022 for scipy >= 1.15 (the solver's unit registry is CODATA 2018: 6.8e-10 apart).
    from scipy.constants import mu_0 as mu_0_scipy

    prefactor = mu_0_scipy / (4 * np.pi) * to_amp_per_meter
    del to_meter
    B = kernels.sheet_field(put(positions), put(areas), put(current_densities), float(z0), put(eval_xyz),
                            prefactor, vector)
    return B.cpu().numpy()


def SheetCurrentField(*, sheet_positions: np.ndarray, current_densities: np.ndarray, z0: float,
                      length_units: str = "um", current_units: str = "uA") -> Parameter:
    """A Parameter giving the z component (tesla) of the field of a 2D sheet of current
    (``sources/current.py:202-245``)."""
    return Parameter(biot_savart_2d, positions=sheet_positions, current_densities=current_densities, z0=z0,
                     length_units=length_units, current_units=current_units, vector=False)
