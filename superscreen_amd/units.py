"""Minimal physical-unit handling for the solver path.

The reference uses a global ``pint.UnitRegistry`` (``units.py:1-3``) for exactly three things on
the hot path: ``field_conversion_factor`` (``solver/utils.py:407-437``), the vortex flux
``Phi_0 / mu_0`` (``solver/solve.py:441-442``) and the fluxoid unit conversions
(``solution.py:538,560``).  pint is not a dependency here; this module implements the small
subset needed: SI-prefixed base units, ``*``, ``/``, ``**`` expressions, and the H <-> B = mu_0 H
convention of ``convert_field`` (``solver/utils.py:350-404``).

Constants are the CODATA 2018 values shipped with pint's default registry.
"""
from __future__ import annotations

import ast
import math
import operator
from typing import Optional, Tuple, Union

import numpy as np

MU_0 = 1.25663706212e-6           # N / A^2
PHI_0 = 2.067833848461929e-15     # Wb  ( = h / (2 e) )

# dimension vector: (length, mass, time, current)
_DIMLESS = (0, 0, 0, 0)


class Unit:
    """A scale factor to SI and a dimension vector."""

    __slots__ = ("scale", "dims", "name")

    def __init__(self, scale: float, dims: Tuple[int, ...], name: str = ""):
        self.scale = float(scale)
        self.dims = tuple(dims)
        self.name = name

    def __mul__(self, other):
        if isinstance(other, Unit):
            return Unit(self.scale * other.scale, tuple(a + b for a, b in zip(self.dims, other.dims)))
        return Unit(self.scale * float(other), self.dims)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return Unit(self.scale / other.scale, tuple(a - b for a, b in zip(self.dims, other.dims)))
        return Unit(self.scale / float(other), self.dims)

    def __rtruediv__(self, other):
        return Unit(float(other) / self.scale, tuple(-a for a in self.dims))

    def __pow__(self, p):
        return Unit(self.scale ** p, tuple(int(a * p) for a in self.dims))

    def __repr__(self):
        return f"Unit({self.name or self.scale!r}, dims={self.dims})"


_PREFIX = {"": 1.0, "T": 1e12, "G": 1e9, "M": 1e6, "k": 1e3, "c": 1e-2, "m": 1e-3, "u": 1e-6,
           "µ": 1e-6, "n": 1e-9, "p": 1e-12, "f": 1e-15}
_LONG_PREFIX = {"tera": 1e12, "giga": 1e9, "mega": 1e6, "kilo": 1e3, "centi": 1e-2,
                "milli": 1e-3, "micro": 1e-6, "nano": 1e-9, "pico": 1e-12, "femto": 1e-15}

_L, _M, _T, _I = (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)
_TESLA = (0, 1, -2, -1)          # kg s^-2 A^-1
_WEBER = (2, 1, -2, -1)
_HENRY = (2, 1, -2, -2)

# symbol -> (scale, dims, prefixable)
_BASE = {
    "m": (1.0, _L, True), "meter": (1.0, _L, True), "metre": (1.0, _L, True),
    "A": (1.0, _I, True), "amp": (1.0, _I, True), "ampere": (1.0, _I, True),
    "T": (1.0, _TESLA, True), "tesla": (1.0, _TESLA, True),
    "G": (1e-4, _TESLA, True), "gauss": (1e-4, _TESLA, True),
    "Wb": (1.0, _WEBER, True), "weber": (1.0, _WEBER, True),
    "H": (1.0, _HENRY, True), "henry": (1.0, _HENRY, True),
    "Oe": (1e3 / (4 * math.pi), (-1, 0, 0, 1), True), "oersted": (1e3 / (4 * math.pi), (-1, 0, 0, 1), True),
    "s": (1.0, _T, True), "kg": (1.0, _M, False), "g": (1e-3, _M, True),
    "Phi_0": (PHI_0, _WEBER, False), "Phi0": (PHI_0, _WEBER, False),
    "magnetic_flux_quantum": (PHI_0, _WEBER, False),
    "mu_0": (MU_0, (1, 1, -2, -2), False), "mu0": (MU_0, (1, 1, -2, -2), False),
    "dimensionless": (1.0, _DIMLESS, False),
}


def _lookup(symbol: str) -> Unit:
    if symbol in _BASE:
        s, d, _ = _BASE[symbol]
        return Unit(s, d, symbol)
    for pre, f in _LONG_PREFIX.items():
        if symbol.startswith(pre) and symbol[len(pre):] in _BASE and _BASE[symbol[len(pre):]][2]:
            s, d, _ = _BASE[symbol[len(pre):]]
            return Unit(f * s, d, symbol)
    pre, rest = symbol[:1], symbol[1:]
    if pre in _PREFIX and rest in _BASE and _BASE[rest][2]:
        s, d, _ = _BASE[rest]
        return Unit(_PREFIX[pre] * s, d, symbol)
    raise ValueError(f"Unknown unit {symbol!r}.")


_BINOPS = {ast.Mult: operator.mul, ast.Div: operator.truediv, ast.Pow: operator.pow}


def _eval(node):
    if isinstance(node, ast.Expression):
        return _eval(node.body)
    if isinstance(node, ast.Name):
        return _lookup(node.id)
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return node.value
    if isinstance(node, ast.BinOp) and type(node.op) in _BINOPS:
        return _BINOPS[type(node.op)](_eval(node.left), _eval(node.right))
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
        return -_eval(node.operand)
    raise ValueError("Unsupported unit expression.")


def parse_units(expr: Union[str, Unit]) -> Unit:
    """Parses e.g. ``"mT"``, ``"uA / um"``, ``"mT * um**2"``, ``"Phi_0"``."""
    if isinstance(expr, Unit):
        return expr
    text = expr.replace("^", "**").replace("µ", "u").strip()
    out = _eval(ast.parse(text, mode="eval"))
    if not isinstance(out, Unit):
        out = Unit(float(out), _DIMLESS)
    out.name = expr
    return out


class Quantity:
    """A magnitude (float or ndarray) with units; the small subset of ``pint.Quantity`` that
    callers of ``polygon_fluxoid`` use (``.magnitude``, ``.units``, ``.to``, float())."""

    __slots__ = ("magnitude", "units")

    def __init__(self, magnitude, units: Union[str, Unit]):
        self.magnitude = magnitude
        self.units = parse_units(units)

    @property
    def m(self):
        return self.magnitude

    def to(self, units: Union[str, Unit]) -> "Quantity":
        new = parse_units(units)
        if new.dims != self.units.dims:
            raise ValueError(f"Cannot convert {self.units.name!r} to {new.name!r}.")
        return Quantity(self.magnitude * (self.units.scale / new.scale), new)

    def __float__(self):
        return float(self.magnitude)

    def __add__(self, other: "Quantity"):
        return Quantity(self.magnitude + other.to(self.units).magnitude, self.units)

    def __repr__(self):
        return f"<Quantity({self.magnitude}, {self.units.name!r})>"


_H_DIMS = (-1, 0, 0, 1)


def convert_field(value, new_units: Union[str, Unit], old_units: Optional[Union[str, Unit]] = None,
                  with_units: bool = True):
    """``convert_field`` (solver/utils.py:350-404): converts between field units, treating
    H ([current]/[length]) and B = mu_0 H as interchangeable."""
    if isinstance(value, Quantity):
        old, mag = value.units, value.magnitude
    else:
        if old_units is None:
            raise ValueError("Old units must be specified if value is not a Quantity.")
        old, mag = parse_units(old_units), value
    new = parse_units(new_units)
    si = np.asarray(mag, dtype=float) * old.scale
    if new.dims != old.dims:
        if old.dims == _H_DIMS and new.dims == _TESLA:
            si = si * MU_0
        elif old.dims == _TESLA and new.dims == _H_DIMS:
            si = si / MU_0
        else:
            raise ValueError(f"Cannot convert {old.name!r} to {new.name!r}.")
    out = si / new.scale
    if out.ndim == 0:
        out = float(out)
    return Quantity(out, new) if with_units else out


def field_conversion_factor(field_units: str, current_units: str, length_units: str = "m") -> float:
    """``field_conversion_factor`` (solver/utils.py:407-437) as a plain float: multiply a field
    given in ``field_units`` (H or B = mu_0 H) by it to get ``current_units / length_units``."""
    return float(convert_field(1.0, f"{current_units} / {length_units}", old_units=field_units,
                               with_units=False))


def vortex_flux(current_units: str, length_units: str) -> float:
    """``ureg("Phi_0 / mu_0").to(current_units * length_units)`` (solver/solve.py:441-442)."""
    target = parse_units(f"{current_units} * {length_units}")
    return PHI_0 / MU_0 / target.scale


def current_to_float(value, current_units: str) -> float:
    """``current_to_float`` (solver/utils.py:327-335): floats pass through, strings such as
    ``"1 mA"`` and Quantities are converted."""
    if isinstance(value, str):
        parts = value.strip().split(None, 1)
        value = Quantity(float(parts[0]), parts[1] if len(parts) > 1 else "dimensionless")
    if isinstance(value, Quantity):
        return float(value.to(current_units).magnitude)
    return value
