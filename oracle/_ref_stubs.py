"""TEST INFRASTRUCTURE ONLY (never imported by the product package).

Makes the *unmodified* reference (`/root/reference/superscreen`, v0.13.0) importable in
THIS container, where numba / pint / h5py / shapely / meshpy / IPython are absent, by
pre-inserting inert stand-in modules into ``sys.modules`` (SURVEY.md §8c "stub recipe").
The numba kernels then run as plain Python loops (``njit`` = identity, ``prange`` =
``range``), i.e. the reference's own arithmetic, executed by CPython.

Nothing from the reference is copied; this file only exists so that
``oracle/make_golden.py`` can *call* the reference and record its outputs as fixtures.
The reference does not exist on the GPU box, so nothing outside ``make_golden.py`` may import
this module.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SUPERSCREEN_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "superscreen"))


def _module(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def install():
    """Install the stand-in modules and put the reference on ``sys.path``."""
    if "superscreen" in sys.modules:
        return sys.modules["superscreen"]

    # numba: njit(...) -> identity decorator, prange -> range
    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    _module("numba", njit=njit, prange=range, set_num_threads=lambda n: None)

    # h5py
    _module("h5py", Group=object, File=object, SoftLink=object)

    # pint
    class _Placeholder:
        def __init__(self, *a, **k):
            pass

    class DimensionalityError(Exception):
        pass

    _module(
        "pint",
        UnitRegistry=_Placeholder,
        Quantity=_Placeholder,
        Unit=_Placeholder,
        DimensionalityError=DimensionalityError,
    )

    # IPython
    ip = _module("IPython")
    ip.display = _module("IPython.display", HTML=_Placeholder)

    # meshpy
    mp = _module("meshpy")
    mp.triangle = _module("meshpy.triangle")

    # shapely
    class _Geom:
        def __init__(self, *a, **k):
            pass

    sh = _module("shapely")
    geo = _module("shapely.geometry", Polygon=_Geom, LinearRing=_Geom, LineString=_Geom,
                  MultiLineString=_Geom, JOIN_STYLE=types.SimpleNamespace(round=1, mitre=2, bevel=3))
    geo.polygon = _module("shapely.geometry.polygon", Polygon=_Geom, LinearRing=_Geom,
                          orient=lambda p, *a, **k: p)
    geo.linestring = _module("shapely.geometry.linestring", LineString=_Geom)
    sh.geometry = geo
    sh.affinity = _module("shapely.affinity")
    sh.ops = _module("shapely.ops", polygonize=lambda *a, **k: [], unary_union=lambda *a, **k: None)
    sh.validation = _module("shapely.validation", explain_validity=lambda *a, **k: "")

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import superscreen  # noqa: F401  (the reference, unmodified)

    return superscreen
