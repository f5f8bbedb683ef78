"""Applied-field sources.  Only the uniform field is on the BASELINE path
(``sources/constant.py:7-32``); dipoles, Pearl vortices and current sheets are out of scope
(SURVEY.md section 2, row 15)."""
from __future__ import annotations

import numpy as np

from .parameter import Parameter


def constant(x, y, z, value=0):
    """Constant field (``sources/constant.py:7-20``)."""
    return value * np.ones_like(x, dtype=float)


def ConstantField(value: float = 0) -> Parameter:
    """A Parameter returning ``value`` at all ``x, y, z`` (``sources/constant.py:23-32``)."""
    return Parameter(constant, value=float(value))
