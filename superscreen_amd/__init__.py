"""superscreen_amd -- MI355X-native hot path of SuperScreen's London-equation solver.

Drop-in for the ``Device`` / ``Layer`` / ``Polygon`` / ``factorize_model`` / ``solve`` path of
loganbvh/superscreen (reference v0.13.0); the numerics run in hand-written HIP kernels for
gfx950 behind the C ABI declared in ``include/superscreen_hip.h``.  See DESIGN.md.
"""
# Process-global HIP settings are the application's to choose, not this package's: nothing is changed at import.
# One that matters for stacks of >= 3 films: HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES
# (default 4) hardware queues per priority level and the factorization schedules keep one high-priority chain
# stream (and, for >= 3 films, one update stream) per film busy while the trailing matrices are large
# (csrc/chol.hip, lu.hip); a four-film stack factors 3 % faster with GPU_MAX_HW_QUEUES=8 set BEFORE the HIP
# runtime starts (bench.py does that).  Do NOT go above 8: with 16 or 32 the streams of a four-film factorization
# take more hardware queues than the chip has queue slots, and every later kernel of the process -- on any stream --
# runs 20-40 % slower (measured, DESIGN_HISTORY.md, round 4).

from .version import __version__  # noqa: F401

_LAZY = {
    "Device": "device", "Layer": "device", "Polygon": "device",
    "Mesh": "mesh", "MeshOperators": "mesh",
    "Parameter": "parameter", "Constant": "parameter",
    "ConstantField": "sources",
    "solve": "solver", "factorize_model": "solver", "FactorizedModel": "solver",
    "LinearSystem": "solver", "FilmInfo": "solver", "LambdaInfo": "solver",
    "convert_field": "units", "field_conversion_factor": "units",
    "Solution": "solution", "FilmSolution": "solution", "Fluxoid": "solution",
    "Vortex": "solution",
    "find_fluxoid_solution": "fluxoid", "make_fluxoid_polygons": "fluxoid",
    "solve_sweep": "sweep",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod = importlib.import_module(f".{_LAZY[name]}", __name__)
        return getattr(mod, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
