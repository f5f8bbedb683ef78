"""Position-dependent parameters (applied fields, penetration depths).

Mirrors the call convention of the reference's ``Parameter`` (``parameter.py:66-133``): a
callable ``f(x, y[, z], **kwargs)`` bound to keyword arguments, evaluated on arrays of
coordinates; arithmetic between Parameters/real numbers yields a ``CompositeParameter``
(``parameter.py:180-317``).  ``Constant`` is ``parameter.py:320-339``.
"""
from __future__ import annotations

import inspect
import numbers
import operator
from typing import Callable

import numpy as np


class Parameter:
    """A callable computing a scalar field as a function of ``x, y`` (and optionally ``z``)."""

    __slots__ = ("func", "kwargs", "_takes_z")

    def __init__(self, func: Callable, **kwargs):
        spec = inspect.getfullargspec(func)
        args = spec.args
        if args[:2] != ["x", "y"]:
            raise ValueError(
                f"The first function arguments must be x and y, not {', '.join(args[:2])!r}."
            )
        nargs = 2
        if "z" in args:
            if args.index("z") != 2:
                raise ValueError(
                    "If the function takes an argument z, it must be the third argument (x, y, z)."
                )
            nargs = 3
        defaults = spec.defaults or []
        if len(defaults) != len(args) - nargs:
            raise ValueError("All arguments other than x, y, z must be keyword arguments.")
        allowed = set(args[nargs:]) | set(spec.kwonlyargs or [])
        if not set(kwargs).issubset(allowed):
            raise ValueError(
                f"Provided keyword arguments {sorted(set(kwargs) - allowed)!r} do not match "
                "the function signature."
            )
        self.func = func
        self.kwargs = dict(zip(args[nargs:], defaults))
        self.kwargs.update(spec.kwonlydefaults or {})
        self.kwargs.update(kwargs)
        self._takes_z = nargs == 3

    def __call__(self, x, y, z=None):
        kwargs = dict(self.kwargs)
        x, y = np.atleast_1d(np.squeeze(x), np.squeeze(y))
        if z is not None and self._takes_z:
            kwargs["z"] = np.atleast_1d(np.squeeze(z))
        result = np.asarray(self.func(x, y, **kwargs)).squeeze()
        if result.ndim == 0:
            result = result.item()
        return result

    def __add__(self, other):
        return CompositeParameter(self, other, operator.add)

    def __radd__(self, other):
        return CompositeParameter(other, self, operator.add)

    def __sub__(self, other):
        return CompositeParameter(self, other, operator.sub)

    def __rsub__(self, other):
        return CompositeParameter(other, self, operator.sub)

    def __mul__(self, other):
        return CompositeParameter(self, other, operator.mul)

    def __rmul__(self, other):
        return CompositeParameter(other, self, operator.mul)

    def __truediv__(self, other):
        return CompositeParameter(self, other, operator.truediv)

    def __rtruediv__(self, other):
        return CompositeParameter(other, self, operator.truediv)

    def __pow__(self, other):
        return CompositeParameter(self, other, operator.pow)

    def __eq__(self, other):
        if other is self:
            return True
        if not isinstance(other, Parameter) or isinstance(other, CompositeParameter):
            return False
        return self.func.__code__ == other.func.__code__ and self.kwargs == other.kwargs

    def __hash__(self):
        return id(self)

    def __repr__(self):
        kw = ", ".join(f"{k}={v!r}" for k, v in self.kwargs.items())
        return f"Parameter<{self.func.__name__}({kw})>"


class CompositeParameter(Parameter):
    """Result of ``+ - * / **`` between Parameters and/or real numbers."""

    __slots__ = ("left", "right", "operator")

    def __init__(self, left, right, op):
        for operand in (left, right):
            if not isinstance(operand, (numbers.Real, Parameter)):
                raise TypeError(f"Unsupported operand type {type(operand)} for Parameter arithmetic.")
        self.left, self.right, self.operator = left, right, op

    def __call__(self, x, y, z=None):
        vals = [v(x, y, z) if isinstance(v, Parameter) else v for v in (self.left, self.right)]
        return self.operator(*vals)

    def __eq__(self, other):
        return (isinstance(other, CompositeParameter) and self.left == other.left
                and self.right == other.right and self.operator is other.operator)

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return f"CompositeParameter<{self.left!r} {self.operator.__name__} {self.right!r}>"


def _constant2d(x, y, value=0):
    return value * np.ones_like(x, dtype=float)


class Constant(Parameter):
    """A Parameter whose value does not depend on position (``parameter.py:320-339``)."""

    __slots__ = ()

    def __init__(self, value, dimensions: int = 2):
        if dimensions not in (2, 3):
            raise ValueError(f"Dimensions must be 2 or 3, got {dimensions}.")
        super().__init__(_constant2d, value=value)
