"""``Layer``, ``Polygon`` and ``Device``: the model objects the solver path consumes.

API surface kept from the reference: ``Layer(name, Lambda=None,
london_lambda=None, thickness=None, z0=0)`` (``device/layer.py:32-64``), ``Polygon(name=None,
*, layer=None, points)`` with CCW orientation and ``contains_points`` via
``matplotlib.path.Path`` (``device/polygon.py:40-80,138-162``), ``Device(name, *, layers,
films, holes=None, terminals=None, abstract_regions=None, length_units="um",
solve_dtype="float32")`` with ``films / holes / layers / meshes / terminals``,
``solve_dtype``, ``length_units``, ``holes_by_film``, ``polygons_by_layer``, ``copy``
(``device/device.py:47-246``).

Out of scope: geometry authoring (boolean ops, buffering, resampling --
needs shapely), meshing with meshpy/Triangle (``Device.make_mesh`` here meshes only what the
synthetic mesher supports, or takes explicit triangulations), plotting, HDF5, transforms.
``mutual_inductance_matrix`` (``device/device.py:538-648``, row 1 of "next" in section 8f) is
implemented: it only loops ``solve(model=...)``.
"""
from __future__ import annotations

import logging
from copy import deepcopy
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from .geometry import close_curve
from .parameter import Parameter


logger = logging.getLogger(__name__)


class Layer:
    """A single layer of a superconducting device (``device/layer.py:11-64``)."""

    __slots__ = ("name", "thickness", "london_lambda", "z0", "_Lambda")

    def __init__(self, name: str, Lambda: Union[float, Parameter, None] = None,
                 london_lambda: Union[float, Parameter, None] = None,
                 thickness: Optional[float] = None, z0: float = 0):
        # exactly one description of the screening length: Lambda itself, or the pair it is derived from
        # (``device/layer.py:40-58``: the same two ValueErrors)
        pair_given = (london_lambda is not None, thickness is not None)
        if Lambda is None and not all(pair_given):
            raise ValueError("You must provide either an effective penetration depth Lambda "
                             "or both a london_lambda and a thickness.")
        if Lambda is not None and any(pair_given):
            raise ValueError("You must provide either an effective penetration depth Lambda "
                             "or both a london_lambda and a thickness (but not all three).")
        self.name, self.z0 = name, z0
        self.london_lambda, self.thickness = london_lambda, thickness
        self._Lambda = Lambda          # None: derived from london_lambda and thickness on access

    @property
    def Lambda(self) -> Union[float, Parameter]:
        """Effective penetration depth ``lambda^2 / d`` (``device/layer.py:60-64``)."""
        if self._Lambda is not None:
            return self._Lambda
        return self.london_lambda ** 2 / self.thickness

    @Lambda.setter
    def Lambda(self, value) -> None:
        if self._Lambda is None:
            raise AttributeError("Can't set Lambda directly. Set london_lambda and/or thickness instead.")
        self._Lambda = value

    def copy(self) -> "Layer":
        return deepcopy(self)

    def to_hdf5(self, h5group) -> None:
        """``device/layer.py:108-116``."""
        from .io import serialize_obj

        h5group.attrs["name"] = self.name
        h5group.attrs["z0"] = self.z0
        if self.thickness is not None:
            h5group.attrs["thickness"] = self.thickness
        if self.london_lambda is not None:
            serialize_obj(h5group, self.london_lambda, "london_lambda", attr=True)
        else:
            serialize_obj(h5group, self.Lambda, "Lambda", attr=True)

    @staticmethod
    def from_hdf5(h5group) -> "Layer":
        """``device/layer.py:118-138``."""
        from .io import deserialize_obj

        Lambda = london_lambda = None
        if "london_lambda" in h5group.attrs or "london_lambda.pickle" in h5group.attrs:
            london_lambda = deserialize_obj(h5group, "london_lambda", attr=True)
        else:
            Lambda = deserialize_obj(h5group, "Lambda", attr=True)
        return Layer(h5group.attrs["name"], Lambda=Lambda, london_lambda=london_lambda,
                     thickness=h5group.attrs.get("thickness", None), z0=h5group.attrs["z0"])

    def __eq__(self, other) -> bool:
        if other is self:
            return True
        if not isinstance(other, Layer):
            return False
        return (self.name == other.name and self.thickness == other.thickness
                and self.london_lambda == other.london_lambda and self.Lambda == other.Lambda
                and self.z0 == other.z0)

    def __repr__(self) -> str:
        return (f"Layer({self.name!r}, Lambda={self.Lambda!r}, thickness={self.thickness!r}, "
                f"london_lambda={self.london_lambda!r}, z0={self.z0!r})")


def _signed_area(points: np.ndarray) -> float:
    x, y = points[:, 0], points[:, 1]
    return 0.5 * float(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))


class Polygon:
    """A simply-connected polygon located in a Layer (``device/polygon.py:28-162``).

    ``points`` are stored closed and counter-clockwise, as the reference does through
    ``shapely.geometry.polygon.orient`` (``device/polygon.py:56-80``).
    """

    __slots__ = ("name", "layer", "_points")

    def __init__(self, name: Optional[str] = None, *, layer: Optional[str] = None, points):
        self.name = name
        self.layer = layer
        self.points = points

    @property
    def points(self) -> np.ndarray:
        return self._points

    @points.setter
    def points(self, points) -> None:
        if isinstance(points, Polygon):
            points = points.points
        points = np.asarray(points, dtype=float)
        if points.ndim != 2 or points.shape[-1] != 2:
            raise ValueError(f"Expected shape (n, 2), but got {points.shape}.")
        points = close_curve(points)
        if len(points) < 4:
            raise ValueError("The given points do not define a valid polygon (fewer than 3 vertices).")
        area = _signed_area(points)
        if area == 0:
            raise ValueError("The given points do not define a valid polygon (zero area).")
        if area < 0:
            points = points[::-1].copy()
        self._points = points

    @property
    def is_valid(self) -> bool:
        return self.name is not None and self.layer is not None and abs(self.area) > 0

    @property
    def area(self) -> float:
        return abs(_signed_area(self._points))

    @property
    def extents(self) -> Tuple[float, float]:
        return tuple(np.ptp(self._points, axis=0))

    @property
    def path(self):
        """A ``matplotlib.path.Path`` of the boundary (``device/polygon.py:108-111``)."""
        from matplotlib.path import Path

        return Path(self._points, closed=True)

    def set_name(self, name) -> "Polygon":
        self.name = name
        return self

    def set_layer(self, layer) -> "Polygon":
        self.layer = layer
        return self

    def contains_points(self, points: np.ndarray, index: bool = False, radius: float = 0):
        """``device/polygon.py:138-162`` -- point-in-polygon via ``Path.contains_points``."""
        mask = self.path.contains_points(np.atleast_2d(points), radius=radius)
        if index:
            return np.where(mask)[0]
        return mask

    # -- affine transforms (``device/polygon.py:226-300``; shapely.affinity there, plain numpy here) --
    def _origin(self, origin) -> np.ndarray:
        if isinstance(origin, str):
            if origin == "center":       # centre of the bounding box
                pts = self.points
                return 0.5 * (pts.min(axis=0) + pts.max(axis=0))
            if origin == "centroid":     # centre of mass of the polygon
                x, y = self.points[:, 0], self.points[:, 1]
                cross = x[:-1] * y[1:] - x[1:] * y[:-1]
                return np.array([np.sum((x[:-1] + x[1:]) * cross), np.sum((y[:-1] + y[1:]) * cross)]) \
                    / (3.0 * np.sum(cross))
            raise ValueError(f"Unknown origin {origin!r}; expected (x, y), 'center' or 'centroid'.")
        return np.asarray(origin, dtype=float)

    def rotate(self, degrees: float, origin=(0.0, 0.0), inplace: bool = False) -> "Polygon":
        """Rotates the polygon counterclockwise by ``degrees`` about ``origin``."""
        polygon = self if inplace else self.copy()
        o = self._origin(origin)
        c, s_ = np.cos(np.radians(degrees)), np.sin(np.radians(degrees))
        polygon.points = (self.points - o) @ np.array([[c, s_], [-s_, c]]) + o
        return polygon

    def translate(self, dx: float = 0.0, dy: float = 0.0, inplace: bool = False) -> "Polygon":
        polygon = self if inplace else self.copy()
        polygon.points = self.points + np.array([dx, dy], dtype=float)
        return polygon

    def scale(self, xfact: float = 1.0, yfact: float = 1.0, origin=(0, 0), inplace: bool = False) -> "Polygon":
        """Scales by ``xfact`` / ``yfact`` about ``origin``; negative factors mirror the polygon."""
        polygon = self if inplace else self.copy()
        o = self._origin(origin)
        polygon.points = (self.points - o) * np.array([xfact, yfact], dtype=float) + o
        return polygon

    def on_boundary(self, points: np.ndarray, radius: float = 1e-3, index: bool = False):
        """Points within ``radius`` of the polygon's boundary (``device/polygon.py:164-190``)."""
        points = np.atleast_2d(points)
        boundary = np.logical_and(self.contains_points(points, radius=radius),
                                  ~self.contains_points(points, radius=-radius))
        return np.where(boundary)[0] if index else boundary

    def to_hdf5(self, h5group) -> None:
        """``device/polygon.py:621-626``."""
        if self.name:
            h5group.attrs["name"] = self.name
        if self.layer:
            h5group.attrs["layer"] = self.layer
        h5group["points"] = self.points

    @staticmethod
    def from_hdf5(h5group) -> "Polygon":
        """``device/polygon.py:628-634``."""
        return Polygon(name=h5group.attrs.get("name", None), layer=h5group.attrs.get("layer", None),
                       points=np.asarray(h5group["points"]))

    def copy(self) -> "Polygon":
        return Polygon(self.name, layer=self.layer, points=self._points.copy())

    def __eq__(self, other) -> bool:
        if other is self:
            return True
        if not isinstance(other, Polygon):
            return False
        return (self.name == other.name and self.layer == other.layer
                and np.allclose(self.points, other.points))

    def __repr__(self) -> str:
        return f"Polygon({self.name!r}, layer={self.layer!r}, points=<ndarray: shape={self._points.shape}>)"


class Device:
    """A device composed of one or more layers of thin-film superconductor
    (``device/device.py:29-109``)."""

    def __init__(self, name: str, *, layers, films, holes=None, terminals=None,
                 abstract_regions=None, length_units: str = "um",
                 solve_dtype: Union[str, np.dtype] = "float32"):
        self.name = name
        if isinstance(layers, dict):
            layers = list(layers.values())
        self.layers: Dict[str, Layer] = {layer.name: layer for layer in layers}
        if isinstance(films, dict):
            films = list(films.values())
        self.films: Dict[str, Polygon] = {film.name: film for film in films}
        if holes is None:
            holes = []
        if isinstance(holes, dict):
            holes = list(holes.values())
        self.holes: Dict[str, Polygon] = {hole.name: hole for hole in holes}
        self.terminals: Dict[str, List[Polygon]] = terminals or {}
        for film, terms in self.terminals.items():  # device/device.py:82-84
            for terminal in terms:
                if film in self.films:
                    terminal.layer = self.films[film].layer
        if not set(self.terminals).issubset(self.films):
            raise ValueError(
                f"terminals.keys() must be a subset of films.keys() ({list(self.films)!r})."
            )
        if abstract_regions is None:
            abstract_regions = []
        if isinstance(abstract_regions, dict):
            abstract_regions = list(abstract_regions.values())
        self.abstract_regions = {region.name: region for region in abstract_regions}
        for polygons, label in [(self.films.values(), "film"), (self.holes.values(), "hole")]:
            for polygon in polygons:
                if not polygon.is_valid:
                    raise ValueError(f"The following {label} is not valid: {polygon}.")
                if polygon.layer not in self.layers:
                    raise ValueError(
                        f"The following {label} is assigned to a layer that doesn not "
                        f"exist in the device: {polygon}."
                    )
        self._length_units = length_units
        self.solve_dtype = solve_dtype
        self.meshes = None

    @property
    def length_units(self) -> str:
        return self._length_units

    @property
    def solve_dtype(self) -> np.dtype:
        """Numpy dtype used for the solve (``device/device.py:116-127``)."""
        return self._solve_dtype

    @solve_dtype.setter
    def solve_dtype(self, dtype) -> None:
        try:
            np.finfo(dtype)
        except ValueError as e:
            raise ValueError(f"Invalid float dtype: {dtype}") from e
        self._solve_dtype = np.dtype(dtype)

    def get_polygons(self, include_terminals: bool = True) -> List[Polygon]:
        polygons = []
        for attr in ("films", "holes", "abstract_regions"):
            polygons.extend(getattr(self, attr).values())
        if include_terminals:
            for terms in self.terminals.values():
                polygons.extend(terms)
        return polygons

    def polygons_by_layer(self, polygon_type: Optional[str] = None) -> Dict[str, List[Polygon]]:
        """``device/device.py:155-196``."""
        valid = ("film", "hole", "abstract", "terminal", "all")
        polygon_type = (polygon_type or "all").lower()
        if polygon_type not in valid:
            raise ValueError(f"Invalid polygon type ({polygon_type}). Expected one of {valid!r}.")
        if polygon_type == "film":
            polys = list(self.films.values())
        elif polygon_type == "hole":
            polys = list(self.holes.values())
        elif polygon_type == "abstract":
            polys = list(self.abstract_regions.values())
        elif polygon_type == "terminal":
            polys = [t for terms in self.terminals.values() for t in terms]
        else:
            polys = self.get_polygons()
        return {layer: [p for p in polys if p.layer == layer] for layer in self.layers}

    def holes_by_film(self) -> Dict[str, List[Polygon]]:
        """``device/device.py:198-211``: holes of the film's layer whose points all lie in it."""
        by_layer = self.polygons_by_layer("hole")
        out = {}
        for film in self.films.values():
            out[film.name] = [h for h in by_layer[film.layer] if film.contains_points(h.points).all()]
        return out

    def copy(self, with_mesh: bool = True, copy_mesh: bool = False) -> "Device":
        """``device/device.py:213-246``.  Quirk kept: the copy does NOT carry ``solve_dtype``
        (it falls back to the default float32), exactly like the reference (:232-240)."""
        device = Device(
            self.name,
            layers=[layer.copy() for layer in self.layers.values()],
            films=[film.copy() for film in self.films.values()],
            holes=[hole.copy() for hole in self.holes.values()],
            terminals={f: [t.copy() for t in ts] for f, ts in self.terminals.items()},
            abstract_regions=[r.copy() for r in self.abstract_regions.values()],
            length_units=self.length_units,
        )
        if with_mesh and self.meshes is not None:
            meshes = self.meshes
            if copy_mesh:
                meshes = {name: mesh.copy() for name, mesh in meshes.items()}
            device.meshes = meshes
        return device

    def __copy__(self):
        return self.copy(with_mesh=True, copy_mesh=False)

    def __deepcopy__(self, memo):
        return self.copy(with_mesh=True, copy_mesh=True)

    def make_mesh(self, triangulations: Optional[Dict[str, Tuple[np.ndarray, np.ndarray]]] = None,
                  **unsupported) -> None:
        """Attaches a mesh to every film.

        The reference meshes with meshpy/Triangle (``device/device.py:383-471``), which is out
        of scope; here the caller supplies ``{film_name: (sites, elements)}`` (e.g. from
        :mod:`superscreen_amd.synthetic` or any external mesher).
        """
        from .mesh import Mesh

        if triangulations is None:
            raise NotImplementedError(
                "superscreen_amd does not ship a mesher (meshpy/Triangle is out of scope): pass "
                "triangulations={film: (sites, elements)} or use superscreen_amd.synthetic."
            )
        if unsupported:
            raise TypeError(f"Unsupported make_mesh arguments: {sorted(unsupported)}")
        missing = set(self.films) - set(triangulations)
        if missing:
            raise ValueError(f"No triangulation given for films {sorted(missing)!r}.")
        self.meshes = {name: Mesh.from_triangulation(*triangulations[name]) for name in self.films}

    def boundary_vertices(self, film: str) -> Optional[np.ndarray]:
        """Boundary vertex indices of a film's mesh, ordered counter-clockwise and rolled so that
        the sequence does not wrap around inside a terminal (``device/device.py:473-500``)."""
        from . import fem

        if self.meshes is None:
            return None
        mesh = self.meshes[film]
        points = mesh.sites
        indices = fem.boundary_vertices(points, mesh.elements)
        if film not in self.terminals:
            return indices
        for terminal in self.terminals[film]:
            terminal_indices = terminal.contains_points(points[indices], index=True)
            discont = np.diff(terminal_indices) != 1
            if np.any(discont):
                i_discont = np.where(discont)[0][0]
                indices = np.roll(indices, -(i_discont + 1))
                break
        return indices

    def mutual_inductance_matrix(self, hole_polygon_mapping: Optional[Dict[str, np.ndarray]] = None,
                                 units: str = "pH", all_iterations: bool = False, progress_bar: bool = False,
                                 **solve_kwargs):
        """``M[i, j] = fluxoid(polygon S_i around hole i) / I_j`` for a current ``I_j`` circulating
        around hole ``j`` (``device/device.py:538-648``): one factorization; the columns (one warm
        ``solve(model=...)`` per hole in the reference) run together as one multi-column solve
        (:func:`superscreen_amd.solve_sweep`).  Returns a :class:`~superscreen_amd.units.Quantity` holding
        the ``(n_holes, n_holes)`` matrix, or a list of them (one per iterate) if ``all_iterations``."""
        from .fluxoid import make_fluxoid_polygons
        from .solver import factorize_model, solve
        from .units import PHI_0, Quantity, parse_units

        holes = self.holes
        hole_names = list(holes)
        if hole_polygon_mapping is None:
            hole_polygon_mapping = make_fluxoid_polygons(self)
        n_holes = len(hole_polygon_mapping)
        for hole_name, polygon in hole_polygon_mapping.items():
            if hole_name not in holes:
                raise ValueError(f"Hole '{hole_name}' does not exist in the device.")
            if not Polygon(points=polygon).contains_points(holes[hole_name].points).all():
                raise ValueError(f"Hole '{hole_name}' is not completely contained within the given polygon.")
        solve_kwargs = dict(solve_kwargs)
        # bookkeeping default as in the reference (device.py:594); what is SOLVED follows ``solve``'s own
        # default, ``iterations = 0`` (solver/solve.py:303), since the reference forwards ``solve_kwargs``
        iterations = solve_kwargs.get("iterations", 1)
        solve_iterations = solve_kwargs.get("iterations", 0)
        solve_kwargs["progress_bar"] = False
        solve_kwargs.pop("current_units", None)
        I_circ_A = 1e-3  # 1 mA; the magnitude is not important (device.py:597)
        if all_iterations:
            n_iter = 1 if len(self.layers) == 1 else iterations + 1
            solution_slice = slice(None)
        else:
            n_iter = 1
            solution_slice = slice(-1, None)
        mutual = np.zeros((n_iter, n_holes, n_holes))
        films_by_hole = {hole.name: film for film, hs in self.holes_by_film().items() for hole in hs}
        to_units = PHI_0 / I_circ_A / parse_units(units).scale  # (Phi_0 / I) -> `units`
        if parse_units(units).dims != parse_units("H").dims:
            raise ValueError(f"{units!r} is not a unit of inductance.")
        if not hole_names:  # the reference's loop over holes never runs: empty matrices
            result = [Quantity(m, units) for m in mutual]
            return result if all_iterations else result[0]
        model = factorize_model(device=self, current_units="mA", circulating_currents={hole_names[0]: "1 mA"})
        I_circ_val = model.circulating_currents[hole_names[0]]
        plain = set(solve_kwargs) <= {"applied_field", "field_units", "iterations", "progress_bar", "return_solutions"}
        if n_holes >= 2 and plain and not self.terminals and solve_kwargs.get("return_solutions", True):
            # every column of M is a solve with its own circulating current: all of them at once as the
            # columns of one multi-right-hand-side solve (solve_sweep) instead of one solve per hole
            from .sources import ConstantField
            from .sweep import solve_sweep

            field = solve_kwargs.get("applied_field") or ConstantField(0)
            logger.info(f"Evaluating the {n_holes} columns of the {self.name!r} mutual inductance matrix at once.")
            columns = solve_sweep(model, [field] * len(hole_names), field_units=solve_kwargs.get("field_units", "mT"),
                                  iterations=solve_iterations, all_iterations=all_iterations,
                                  circulating_currents=[{name: I_circ_val} for name in hole_names])
        else:
            columns = []
            for j, hole_name in enumerate(hole_names):
                logger.info(f"Evaluating {self.name!r} mutual inductance matrix column "
                            f"({j + 1}/{len(hole_names)}), source = {hole_name!r}.")
                model.set_circulating_currents({hole_name: I_circ_val})
                columns.append(solve(model=model, **solve_kwargs)[solution_slice])
        for j, solutions in enumerate(columns):
            for n, solution in enumerate(solutions):
                for i, name in enumerate(hole_names):
                    fluxoid = solution.polygon_fluxoid(hole_polygon_mapping[name], film=films_by_hole[name],
                                                       units="Phi_0", with_units=False)
                    mutual[n, i, j] = sum(fluxoid) * to_units
        result = [Quantity(m, units) for m in mutual]
        if not all_iterations:
            assert len(result) == 1
            return result[0]
        return result

    # -- rigid transforms of the whole device (``device/device.py:256-381``) ------------------------
    def _warn_if_mesh_exist(self, method: str) -> None:
        if self.meshes:
            logger.warning(f"Calling device.{method} on a device whose mesh already exists returns a new device "
                           f"with no mesh. Call new_device.make_mesh() to generate the mesh for the new device.")

    @staticmethod
    def _check_origin(origin) -> None:
        import numbers

        if not (isinstance(origin, tuple) and len(origin) == 2
                and all(isinstance(val, numbers.Real) for val in origin)):
            raise TypeError("Origin must be a tuple of floats (x, y).")

    def scale(self, xfact: float = 1, yfact: float = 1, origin: Tuple[float, float] = (0, 0)) -> "Device":
        self._check_origin(origin)
        self._warn_if_mesh_exist("scale()")
        device = self.copy(with_mesh=False)
        for polygon in device.get_polygons():
            polygon.scale(xfact=xfact, yfact=yfact, origin=origin, inplace=True)
        return device

    def rotate(self, degrees: float, origin: Tuple[float, float] = (0, 0)) -> "Device":
        self._check_origin(origin)
        self._warn_if_mesh_exist("rotate()")
        device = self.copy(with_mesh=False)
        for polygon in device.get_polygons():
            polygon.rotate(degrees, origin=origin, inplace=True)
        return device

    def mirror_layers(self, about_z: float = 0.0) -> "Device":
        self._warn_if_mesh_exist("mirror_layers()")
        device = self.copy(with_mesh=False)
        for layer in device.layers.values():
            layer.z0 = about_z - layer.z0
        return device

    def translate(self, dx: float = 0, dy: float = 0, dz: float = 0, inplace: bool = False) -> "Device":
        """Moves polygons, mesh sites and (``dz``) layers; the mesh survives (``device/device.py:334-365``).

        What the path keeps per mesh follows the move: the GPU copies of the geometry (``operators._device_cache``: the
        kernels only use coordinate DIFFERENCES inside a film, but the coupling sums between films use both films'
        absolute positions) and the point-in-polygon results (``_contains_cache``) are dropped and rebuilt on next use."""
        device = self if inplace else self.copy(with_mesh=True, copy_mesh=True)
        for polygon in device.get_polygons():
            polygon.translate(dx, dy, inplace=True)
        for mesh in (device.meshes or {}).values():
            mesh.sites += np.array([[dx, dy]], dtype=float)
            mesh.triangle_centroids += np.array([[dx, dy]], dtype=float)
            mesh._triangulation = None
            if dx or dy:
                if mesh.operators is not None:
                    mesh.operators._device_cache.clear()
                mesh.__dict__.pop("_contains_cache", None)
        if dz:
            for layer in device.layers.values():
                layer.z0 += dz
        return device

    def translation(self, dx: float, dy: float, dz: float = 0):
        """Context manager: the device is translated inside the block and moved back afterwards."""
        from contextlib import contextmanager

        @contextmanager
        def moved():
            try:
                self.translate(dx, dy, dz=dz, inplace=True)
                yield
            finally:
                self.translate(-dx, -dy, dz=-dz, inplace=True)

        return moved()

    def to_hdf5(self, path_or_group, save_mesh: bool = True, compress: bool = True) -> None:
        """Serializes the device (``device/device.py:936-977``): same group / attribute names as the
        reference's HDF5 layout, on h5py or on the ``.npz`` container of :mod:`superscreen_amd.io`."""
        from contextlib import nullcontext

        from . import io

        ctx = nullcontext(path_or_group) if io.is_group(path_or_group) else io.open_file(path_or_group, "x")
        with ctx as h5group:
            h5group.attrs["name"] = self.name
            h5group.attrs["length_units"] = self.length_units
            h5group.attrs["solve_dtype"] = str(self.solve_dtype)
            groups = {key: h5group.create_group(key)
                      for key in ("layers", "films", "holes", "terminals", "abstract_regions")}
            for name, layer in self.layers.items():
                layer.to_hdf5(groups["layers"].create_group(name))
            for key, polygons in (("films", self.films), ("holes", self.holes),
                                  ("abstract_regions", self.abstract_regions)):
                for name, polygon in polygons.items():
                    polygon.to_hdf5(groups[key].create_group(name))
            for film_name, terminals in self.terminals.items():
                grp = groups["terminals"].create_group(film_name)
                for i, terminal in enumerate(terminals):
                    terminal.to_hdf5(grp.create_group(str(i)))
            if save_mesh and self.meshes:
                mesh_grp = h5group.create_group("mesh")
                for name, mesh in self.meshes.items():
                    mesh.to_hdf5(mesh_grp.create_group(name), compress=compress)

    @staticmethod
    def from_hdf5(path_or_group) -> "Device":
        """``device/device.py:979-1016``."""
        from contextlib import nullcontext

        from . import io
        from .mesh import Mesh

        ctx = nullcontext(path_or_group) if io.is_group(path_or_group) else io.open_file(path_or_group, "r")
        with ctx as h5group:
            terminals = {}
            for film, grp in h5group["terminals"].items():
                terminals[film] = [Polygon.from_hdf5(grp[str(i)]) for i in range(len(grp))]
            device = Device(
                name=h5group.attrs["name"],
                layers=[Layer.from_hdf5(grp) for grp in h5group["layers"].values()],
                films=[Polygon.from_hdf5(grp) for grp in h5group["films"].values()],
                holes=[Polygon.from_hdf5(grp) for grp in h5group["holes"].values()],
                terminals=terminals or None,
                abstract_regions=[Polygon.from_hdf5(grp) for grp in h5group["abstract_regions"].values()],
                length_units=h5group.attrs["length_units"],
                solve_dtype=h5group.attrs["solve_dtype"],
            )
            if "mesh" in h5group:
                device.meshes = {name: Mesh.from_hdf5(grp) for name, grp in h5group["mesh"].items()}
            return device

    def __eq__(self, other) -> bool:
        """Same name, layers, films, holes, terminals, abstract regions and length units, in any
        order (``device/device.py:1048-1071``); meshes are not compared."""
        if other is self:
            return True
        if not isinstance(other, Device):
            return False

        def same(first, second):
            return sorted(first, key=lambda x: x.name) == sorted(second, key=lambda x: x.name)

        return (self.name == other.name
                and same(self.layers.values(), other.layers.values())
                and same(self.films.values(), other.films.values())
                and same(self.holes.values(), other.holes.values())
                and self.terminals == other.terminals
                and same(self.abstract_regions.values(), other.abstract_regions.values())
                and self.length_units == other.length_units)

    __hash__ = None

    def __repr__(self) -> str:
        return (f"Device({self.name!r}, layers={list(self.layers)!r}, films={list(self.films)!r}, "
                f"holes={list(self.holes)!r}, length_units={self.length_units!r})")
