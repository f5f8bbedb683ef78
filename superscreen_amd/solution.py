"""Result containers and the fluxoid post-processing that parity is judged on.

Kept from the reference (``solution.py``): ``Fluxoid`` (:39-59), ``Vortex`` (:62-92),
``FilmSolution`` (:95-198), ``Solution.__init__`` (:219-244), ``interp_current_density``
(:278-319), ``polygon_fluxoid`` (:484-563), ``hole_fluxoid`` (:565-609).  All of it is O(n)
host work on the vectors the GPU returns.  Field maps at arbitrary positions, vector
potential, HDF5 and plotting are out of scope.
"""
from __future__ import annotations

import datetime as dt
from dataclasses import dataclass
from typing import Callable, Dict, List, NamedTuple, Optional, Union

import numpy as np

from .device import Device, Polygon
from .parameter import Constant
from .units import MU_0, Quantity, convert_field, parse_units
from .version import __version__


def _sum_quantities(values):
    total = values[0]
    for v in values[1:]:
        total = total + v
    return total


class Fluxoid(NamedTuple):
    """Flux part and supercurrent part of the fluxoid of a closed region (``solution.py:39-59``)."""

    flux_part: Union[float, Quantity]
    supercurrent_part: Union[float, Quantity]


@dataclass
class Vortex:
    """A vortex pinned at ``(x, y)`` in ``film`` carrying ``nPhi0`` flux quanta
    (``solution.py:62-92``); handled by the vortex branch of ``solve_film``
    (``solver/solve_film.py:541-554``) as one extra right-hand side per vortex."""

    x: float
    y: float
    film: str
    nPhi0: float = 1

    def to_hdf5(self, h5group) -> None:
        """``solution.py:79-83``."""
        h5group.attrs["x"] = self.x
        h5group.attrs["y"] = self.y
        h5group.attrs["film"] = self.film
        h5group.attrs["nPhi0"] = self.nPhi0

    @staticmethod
    def from_hdf5(h5group) -> "Vortex":
        """``solution.py:85-92``."""
        return Vortex(x=h5group.attrs["x"], y=h5group.attrs["y"], film=h5group.attrs["film"],
                      nPhi0=h5group.attrs["nPhi0"])


class FilmSolution:
    """Raw solution data for a single film (``solution.py:95-130``)."""

    def __init__(self, stream, current_density, applied_field, self_field,
                 field_from_other_films=None):
        self.stream = np.asarray(stream)
        self.current_density = np.asarray(current_density)
        self.applied_field = np.asarray(applied_field)
        self.self_field = np.asarray(self_field)
        if field_from_other_films is not None:
            field_from_other_films = np.asarray(field_from_other_films)
        self.field_from_other_films = field_from_other_films
        self._total_field = None

    @property
    def total_field(self) -> np.ndarray:
        """``applied + self (+ other)`` (``solution.py:123-130``)."""
        if self._total_field is None:
            total = self.applied_field + self.self_field
            if self.field_from_other_films is not None:
                total = total + self.field_from_other_films
            self._total_field = total
        return self._total_field

    def to_hdf5(self, h5group) -> None:
        """``solution.py:132-143``."""
        h5group["stream"] = self.stream
        h5group["current_density"] = self.current_density
        h5group["applied_field"] = self.applied_field
        h5group["self_field"] = self.self_field
        if self.field_from_other_films is not None:
            h5group["field_from_other_films"] = self.field_from_other_films

    @staticmethod
    def from_hdf5(h5group) -> "FilmSolution":
        """``solution.py:145-164``."""
        other = h5group.get("field_from_other_films", None)
        return FilmSolution(stream=np.array(h5group["stream"]), current_density=np.array(h5group["current_density"]),
                            applied_field=np.array(h5group["applied_field"]),
                            self_field=np.array(h5group["self_field"]),
                            field_from_other_films=None if other is None else np.array(other))

    def is_close(self, other: "FilmSolution", rtol: float = 1e-4, atol: float = 1e-7) -> bool:
        """``solution.py:166-185``."""
        kw = dict(rtol=rtol, atol=atol)
        return (np.allclose(self.stream, other.stream, **kw)
                and np.allclose(self.applied_field, other.applied_field, **kw)
                and np.allclose(self.self_field, other.self_field, **kw)
                and np.allclose(self.total_field, other.total_field, **kw))

    def __eq__(self, other) -> bool:
        if other is self:
            return True
        if not isinstance(other, FilmSolution):
            return False
        if (self.field_from_other_films is None) != (other.field_from_other_films is None):
            return False
        return self.is_close(other)


class Solution:
    """Stream functions and fields of all films of a solved ``Device`` (``solution.py:201-244``)."""

    def __init__(self, *, device: Device, film_solutions: Dict[str, FilmSolution],
                 applied_field_func: Callable, field_units: str, current_units: str,
                 circulating_currents: Optional[Dict[str, float]] = None,
                 terminal_currents: Optional[Dict[str, float]] = None,
                 vortices: Optional[List[Vortex]] = None, solver: str = "superscreen_amd.solve",
                 _device_is_copy: bool = False):
        # solution.py:232 -- a sweep builds hundreds of Solutions: it copies the device once and
        # hands the same copy to all of them
        self.device = device if _device_is_copy else device.copy(with_mesh=True, copy_mesh=False)
        self.film_solutions = film_solutions
        self.applied_field_func = applied_field_func
        self.circulating_currents = circulating_currents or {}
        self.terminal_currents = terminal_currents or {}
        self.vortices = vortices or []
        self._field_units = field_units
        self._current_units = current_units
        self._solver = solver
        self._time_created = dt.datetime.now()
        self._version_info = {"superscreen_amd": __version__}

    @property
    def field_units(self) -> str:
        return self._field_units

    @property
    def current_units(self) -> str:
        return self._current_units

    @property
    def solver(self) -> str:
        return self._solver

    @property
    def time_created(self) -> dt.datetime:
        return self._time_created

    @property
    def version_info(self) -> Dict[str, str]:
        return self._version_info

    # ------------------------------------------------------------------------------------
    def interp_current_density(self, positions: np.ndarray, *, film: str, method: str = "linear",
                               units: Optional[str] = None, with_units: bool = False):
        """Interpolates ``J = [dg/dy, -dg/dx]`` inside a film (``solution.py:278-319``);
        positions outside the film or with non-finite interpolants give 0 (:313-315)."""
        import matplotlib.tri as mtri

        device = self.device
        default_units = f"{self.current_units} / {device.length_units}"
        units = units or default_units
        positions = np.atleast_2d(positions)
        xv, yv = positions.T
        interp = {"linear": mtri.LinearTriInterpolator, "cubic": mtri.CubicTriInterpolator}[method]
        mesh = device.meshes[film]
        J = self.film_solutions[film].current_density
        J = np.array([interp(mesh.triangulation, J[:, 0])(xv, yv).data,
                      interp(mesh.triangulation, J[:, 1])(xv, yv).data]).T
        J[~device.films[film].contains_points(positions)] = 0
        J[~np.isfinite(J).all(axis=1)] = 0
        q = Quantity(J, default_units).to(units)
        return q if with_units else q.magnitude

    def current_through_path(self, path_coords: np.ndarray, *, film: str, interp_method: str = "linear",
                             units: Optional[str] = None, with_units: bool = True):
        """Total current crossing a path (``solution.py:321-362``): J at the edge centres, dotted with
        the edge normals ``dr x z`` (``geometry.py:12-29``), times the edge lengths, summed with the
        reference's unit-spacing trapezoid rule."""
        device = self.device
        units = units or self.current_units
        path_coords = np.asarray(path_coords, dtype=float)
        centres = (path_coords[:-1] + path_coords[1:]) / 2
        J_edge = self.interp_current_density(centres, film=film, method=interp_method, with_units=False)
        dr = np.diff(path_coords, axis=0)
        lengths = np.linalg.norm(dr, axis=1)
        normals = np.stack([dr[:, 1], -dr[:, 0]], axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            normals = normals / lengths[:, None]
        J_dot_n = np.sum(J_edge * normals, axis=1)
        total = Quantity(np.trapezoid(J_dot_n * lengths), self.current_units).to(units)
        return total if with_units else total.magnitude

    def to_hdf5(self, path_or_group, device_path: Optional[str] = None, compress: bool = True) -> None:
        """Saves the Solution (``solution.py:936-980``): same layout as the reference's HDF5 files, on
        h5py or on the ``.npz`` container of :mod:`superscreen_amd.io`.  ``device_path``: where in the
        file the device is already stored (a link is written instead of a second copy)."""
        from contextlib import nullcontext

        from . import io

        ctx = nullcontext(path_or_group) if io.is_group(path_or_group) else io.open_file(path_or_group, "x")
        with ctx as h5group:
            h5group.attrs["time_created"] = self.time_created.isoformat()
            h5group.attrs["field_units"] = self.field_units
            h5group.attrs["current_units"] = self.current_units
            h5group.attrs["solver"] = self.solver
            h5group.create_group("version_info").attrs.update(self.version_info)
            if device_path is None:
                self.device.to_hdf5(h5group.create_group("device"), save_mesh=True, compress=compress)
            else:
                h5group["device"] = io.soft_link(h5group, device_path)
            grp = h5group.create_group("film_solutions")
            for name, film_solution in self.film_solutions.items():
                film_solution.to_hdf5(grp.create_group(name))
            vortices_grp = h5group.create_group("vortices")
            for i, vortex in enumerate(self.vortices):
                vortex.to_hdf5(vortices_grp.create_group(str(i)))
            io.serialize_obj(h5group, self.applied_field_func, "applied_field_func")
            h5group.create_group("circulating_currents").attrs.update(self.circulating_currents)
            term_grp = h5group.create_group("terminal_currents")
            for film_name, current_dict in self.terminal_currents.items():
                term_grp.create_group(film_name).attrs.update(current_dict)

    @staticmethod
    def from_hdf5(path_or_group) -> "Solution":
        """``solution.py:982-1030``."""
        from contextlib import nullcontext

        from . import io

        ctx = nullcontext(path_or_group) if io.is_group(path_or_group) else io.open_file(path_or_group, "r")
        with ctx as h5group:
            device = Device.from_hdf5(h5group["device"])
            film_solutions = {name: FilmSolution.from_hdf5(grp) for name, grp in h5group["film_solutions"].items()}
            vortices = [Vortex.from_hdf5(h5group[f"vortices/{i}"]) for i in sorted(h5group["vortices"], key=int)]
            terminal_currents = {film: dict(grp.attrs) for film, grp in h5group["terminal_currents"].items()}
            solution = Solution(
                device=device, film_solutions=film_solutions,
                applied_field_func=io.deserialize_obj(h5group, "applied_field_func"), vortices=vortices,
                circulating_currents=dict(h5group["circulating_currents"].attrs),
                terminal_currents=terminal_currents, current_units=h5group.attrs["current_units"],
                field_units=h5group.attrs["field_units"], solver=h5group.attrs["solver"])
            solution._time_created = dt.datetime.fromisoformat(h5group.attrs["time_created"])
            solution._version_info = dict(h5group["version_info"].attrs)
        return solution

    @staticmethod
    def save_solutions(solutions, path_or_group, compress: bool = True) -> None:
        """A series of Solutions in one file, the device stored once (``solution.py:1032-1064``)."""
        from contextlib import nullcontext

        from . import io

        if not solutions:
            return
        device = solutions[0].device
        ctx = nullcontext(path_or_group) if io.is_group(path_or_group) else io.open_file(path_or_group, "x")
        with ctx as h5group:
            device_grp = h5group.create_group("device")
            device.to_hdf5(device_grp)
            for i, solution in enumerate(solutions):
                device_path = device_grp.name if solution.device == device else None
                solution.to_hdf5(h5group.create_group(str(i)), device_path=device_path, compress=compress)

    @staticmethod
    def load_solutions(path_or_group) -> List["Solution"]:
        """``solution.py:1066-1087``."""
        from contextlib import nullcontext

        from . import io

        ctx = nullcontext(path_or_group) if io.is_group(path_or_group) else io.open_file(path_or_group, "r")
        with ctx as h5group:
            groups = sorted((key for key in h5group if key.isdigit()), key=int)
            return [Solution.from_hdf5(h5group[group]) for group in groups]

    def equals(self, other, require_same_timestamp: bool = False) -> bool:
        """``solution.py:1089-1126``: same device, units, currents, applied field, vortices and film
        solutions (to ``FilmSolution.is_close`` tolerances)."""
        if other is self:
            return True
        if not isinstance(other, Solution):
            return False
        if not (self.device == other.device and self.field_units == other.field_units
                and self.current_units == other.current_units
                and self.circulating_currents == other.circulating_currents
                and getattr(self, "terminal_currents", None) == getattr(other, "terminal_currents", None)
                and self.applied_field_func == other.applied_field_func and self.vortices == other.vortices):
            return False
        if require_same_timestamp and self.time_created != other.time_created:
            return False
        return self.film_solutions == other.film_solutions

    def __eq__(self, other) -> bool:
        return self.equals(other, require_same_timestamp=True)

    __hash__ = None

    def interp_field(self, positions: np.ndarray, *, film: str, dataset: str = "field", method: str = "linear",
                     units: Optional[str] = None, with_units: bool = False):
        """Interpolates the z component of a field inside a film (``solution.py:364-428``)."""
        import matplotlib.tri as mtri

        valid = ("field", "self_field", "applied_field", "field_from_other_films")
        if dataset not in valid:
            raise ValueError(f"Invalid dataset: {dataset!r}. Expected one of {valid!r}")
        if units is None:
            units = self.field_units
        mesh = self.device.meshes[film]
        fs = self.film_solutions[film]
        if dataset == "field":
            field = fs.total_field
        elif dataset == "self_field":
            field = fs.self_field
        elif dataset == "applied_field":
            field = fs.applied_field
        else:
            field = fs.field_from_other_films
            if field is None:
                field = np.zeros(len(mesh.sites))
        interp = {"linear": mtri.LinearTriInterpolator, "cubic": mtri.CubicTriInterpolator}[method]
        positions = np.atleast_2d(positions)
        Hz = interp(mesh.triangulation, field)(positions[:, 0], positions[:, 1]).data
        return convert_field(Hz, units, old_units=self.field_units, with_units=with_units)

    @staticmethod
    def _split_positions(positions, zs, dtype):
        """``(m, 2)`` + ``zs`` or ``(m, 3)`` -> ``(m, 2)``, ``(m,)`` (``solution.py:657-673``)."""
        positions = np.atleast_2d(positions)
        if positions.shape[1] == 3:
            if zs is not None:
                raise ValueError("If positions has shape (m, 3) then zs cannot be specified.")
            zs = positions[:, 2]
            positions = positions[:, :2]
        else:
            zs = np.squeeze(zs)
            if zs.ndim == 0:
                zs = zs.item() * np.ones(positions.shape[0], dtype=dtype)
        if not isinstance(zs, np.ndarray):
            raise ValueError(f"Expected zs to be an ndarray, but got {type(zs)}.")
        return positions, zs

    def screening_field_at_position(self, positions: np.ndarray, *, zs=None, vector: bool = False,
                                    interp_method: str = "linear", units: Optional[str] = None,
                                    with_units: bool = True, return_sum: bool = True):
        """Field of the currents in the device anywhere in space, without the applied field
        (``solution.py:611-723``): inside a film's plane the solved ``self_field`` is interpolated,
        everywhere else the film's sheet current is summed with Biot-Savart (GPU, ``ssa_sheet_field``)."""
        from .sources import biot_savart_2d

        device = self.device
        dtype = device.solve_dtype
        units = units or self.field_units
        positions, zs = self._split_positions(positions, zs, dtype)
        fields = {}
        for name, film in device.films.items():
            layer = device.layers[film.layer]
            field_from_film = np.zeros((len(positions), 3) if vector else len(positions), dtype=dtype)
            in_film = np.zeros(len(positions), dtype=bool)
            if np.all(zs == layer.z0):
                in_film[film.contains_points(positions)] = True
                field_in_film = self.interp_field(positions[in_film], film=film.name, dataset="self_field",
                                                  method=interp_method, units="tesla", with_units=False)
                field_in_film = np.atleast_1d(field_in_film)
                if vector:
                    zeros = np.zeros_like(field_in_film)
                    field_in_film = np.array([zeros, zeros, field_in_film]).T
                field_from_film[in_film] = field_in_film
            out = ~in_film
            if out.any():
                field_from_film[out] = biot_savart_2d(
                    positions[out, 0], positions[out, 1], zs[out], positions=device.meshes[name].sites,
                    areas=device.meshes[name].vertex_areas,
                    current_densities=self.film_solutions[name].current_density, z0=layer.z0,
                    length_units=device.length_units, current_units=self.current_units, vector=vector)
            fields[name] = convert_field(field_from_film, units, old_units="tesla", with_units=with_units)
        if return_sum:
            return sum(fields.values()) if not with_units else _sum_quantities(list(fields.values()))
        return fields

    def field_at_position(self, positions: np.ndarray, *, zs=None, interp_method: str = "linear",
                          units: Optional[str] = None, with_units: bool = True, return_sum: bool = True):
        """Total z field (applied + screening) anywhere in space (``solution.py:725-831``)."""
        device = self.device
        dtype = device.solve_dtype
        units = units or self.field_units
        positions, zs = self._split_positions(positions, zs, dtype)
        fields = self.screening_field_at_position(positions, zs=zs, vector=False, interp_method=interp_method,
                                                  units=self.field_units, with_units=False, return_sum=False)
        films_by_layer = device.polygons_by_layer("film")
        Hz_applied = np.zeros(len(positions), dtype=dtype)
        in_film = np.zeros(len(positions), dtype=bool)
        for name, layer in device.layers.items():
            if np.all(zs == layer.z0):
                for film in films_by_layer[name]:
                    ix = film.contains_points(positions)
                    in_film[ix] = True
                    Hz_applied[ix] = self.interp_field(positions[ix], film=film.name, dataset="applied_field",
                                                       method=interp_method, units=self.field_units)
                    Hz_applied[ix] += self.interp_field(positions[ix], film=film.name,
                                                        dataset="field_from_other_films", method=interp_method,
                                                        units=self.field_units)
                break
        mask = ~in_film
        if mask.any():
            Hz_applied[mask] = np.squeeze(self.applied_field_func(positions[mask, 0], positions[mask, 1],
                                                                  zs[mask, np.newaxis]))
        fields["applied_field"] = np.atleast_1d(Hz_applied).squeeze()
        for key, value in fields.items():
            fields[key] = convert_field(value, units, old_units=self.field_units, with_units=with_units)
        if return_sum:
            return sum(fields.values()) if not with_units else _sum_quantities(list(fields.values()))
        return fields

    def vector_potential_at_position(self, positions: np.ndarray, *, zs=None, units: Optional[str] = None,
                                     with_units: bool = True, return_sum: bool = True):
        """Vector potential of the currents in the device anywhere in space (``solution.py:833-934``):
        ``A(r) = mu_0 / (4 pi) sum_k a_k J_k / |r - r_k|`` per film, shape ``(m, 3)`` with ``A_z = 0``.
        The all-pairs sum runs on the GPU (``ssa_sheet_potential``; cdist + einsum in the reference)."""
        import torch

        from . import _hip, kernels

        _hip.require_gpu()
        device = self.device
        dtype = device.solve_dtype
        units = units or f"{self.field_units} * {device.length_units}"
        positions, zs = self._split_positions(positions, zs, dtype)
        new = parse_units(units)
        length = parse_units(device.length_units).scale
        current = parse_units(self.current_units).scale
        tesla_metre, ampere = (1, 1, -2, -1), (0, 0, 0, 1)
        if new.dims == tesla_metre:
            to_units = MU_0 / (4 * np.pi) * current / new.scale
        elif new.dims == ampere:  # H-like field units: A = mu_0 (...)  ->  (...) in current units
            to_units = 1.0 / (4 * np.pi) * current / new.scale
        else:
            raise ValueError(f"{units!r} is not a unit of vector potential.")
        del length  # (current/length) * length^2 / length = current: no length factor is left
        dev = torch.device("cuda", torch.cuda.current_device())

        def put(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)

        ev = put(np.column_stack([positions, zs]))
        out = {}
        for name, film in device.films.items():
            z0 = device.layers[film.layer].z0
            if np.all(zs - z0 == 0) and film.contains_points(positions).all():
                raise ValueError(f"Cannot evaluate vector potential inside the film ({name!r}).")
            mesh = device.meshes[name]
            Axy = kernels.sheet_potential(put(mesh.sites), put(mesh.vertex_areas),
                                          put(self.film_solutions[name].current_density), float(z0), ev,
                                          to_units).cpu().numpy()
            A = np.concatenate([Axy, np.zeros_like(Axy[:, :1])], axis=1)
            out[name] = Quantity(A, new) if with_units else A
        if return_sum:
            vals = list(out.values())
            return _sum_quantities(vals) if with_units else sum(vals)
        return out

    def polygon_flux(self, name: str, units: Optional[str] = None, with_units: bool = True):
        """Flux through a film, hole or abstract region of the device (``solution.py:430-482``):
        ``sum_i total_field_i w_i`` over the mesh sites inside the polygon."""
        device = self.device
        polygons = {p.name: p for p in device.get_polygons(include_terminals=False)}
        if name not in polygons:
            raise ValueError(f"Unknown polygon: {name!r}.")
        units = units or f"{self.field_units} * {device.length_units}**2"
        polygon = polygons[name]
        if name in device.films:
            film_name = name
        else:
            film_name = None
            for film in device.films.values():
                if film.layer == polygon.layer and film.contains_points(polygon.points).all():
                    film_name = film.name
                    break
            if film_name is None:
                raise ValueError(f"Polygon {name!r} is not contained in any film of its layer.")
        mesh = device.meshes[film_name]
        ix = polygon.contains_points(mesh.sites, index=True)
        total_field = self.film_solutions[film_name].total_field
        flux_raw = float(np.einsum("i, i ->", total_field[ix], mesh.vertex_areas[ix]))
        flux_T_m2 = convert_field(flux_raw, "T", old_units=self.field_units, with_units=False) \
            * parse_units(device.length_units).scale ** 2
        new = parse_units(units)
        if new.dims == (0, 0, 0, 1) or new.dims == (1, 0, 0, 1):  # H-like field units * area
            q = Quantity(flux_T_m2 / MU_0, "A * m").to(units)
        else:
            q = Quantity(flux_T_m2, "Wb").to(units)
        return q if with_units else float(q.magnitude)

    def polygon_fluxoid(self, polygon_coords, *, film: str, interp_method: str = "linear",
                        units: Optional[str] = "Phi_0", with_units: bool = True) -> Fluxoid:
        """Fluxoid of a polygonal region (``solution.py:484-563``):
        flux part ``sum_{i in polygon} total_field_i w_i`` (:535-538) and supercurrent part
        ``mu_0 trapezoid(Lambda_k (J_k . dl_k))`` over the polygon vertices (:542-559)."""
        device = self.device
        if units is None:
            units = f"{self.field_units} * {device.length_units} ** 2"
        polygon = Polygon(points=polygon_coords)
        points = polygon.points
        if not device.films[film].contains_points(points).all():
            raise ValueError(f"The polygon is not contained within the film ({film!r}).")
        mesh = device.meshes[film]
        ix = polygon.contains_points(mesh.sites)
        fields = self.film_solutions[film].total_field
        flux_raw = float(np.einsum("i, i ->", fields[ix], mesh.vertex_areas[ix]))
        # field_units may be H-like or B-like; express as B = mu_0 H, like pint's .to() chain
        flux_T_m2 = convert_field(flux_raw, "T", old_units=self.field_units, with_units=False) \
            * parse_units(device.length_units).scale ** 2
        flux_part = Quantity(flux_T_m2, "Wb").to(units)

        J_units = f"{self.current_units} / {device.length_units}"
        J_poly = self.interp_current_density(points, film=film, method=interp_method,
                                             units=J_units, with_units=False)
        Lambda = device.layers[device.films[film].layer].Lambda
        if not callable(Lambda):
            Lambda = Constant(Lambda)
        Lambda_poly = Lambda(points[:, 0], points[:, 1]) * np.ones(len(points))
        dl = np.diff(points, axis=0)
        int_J = float(np.trapezoid(Lambda_poly[:-1] * np.sum(J_poly[:-1] * dl, axis=1)))
        # [J_units * length^2] = current * length ; mu_0 * that is a flux
        int_J_SI = int_J * parse_units(self.current_units).scale * parse_units(device.length_units).scale
        supercurrent_part = Quantity(MU_0 * int_J_SI, "Wb").to(units)
        if not with_units:
            return Fluxoid(float(flux_part.magnitude), float(supercurrent_part.magnitude))
        return Fluxoid(flux_part, supercurrent_part)

    def hole_fluxoid(self, hole_name: str, points: Optional[np.ndarray] = None,
                     interp_method: str = "linear", units: Optional[str] = "Phi_0",
                     with_units: bool = True) -> Fluxoid:
        """Fluxoid of a polygon enclosing a hole (``solution.py:565-609``).  Without ``points`` the
        polygon comes from :func:`superscreen_amd.fluxoid.make_fluxoid_polygons`."""
        device = self.device
        if points is None:
            from .fluxoid import make_fluxoid_polygons

            points = make_fluxoid_polygons(device, holes=hole_name)[hole_name]
        hole = device.holes[hole_name]
        if not Polygon(points=points).contains_points(hole.points).all():
            raise ValueError(f"Hole {hole.name} is not completely enclosed by the given polygon.")
        film_name = None
        for name, holes in device.holes_by_film().items():
            if hole.name in [h.name for h in holes]:
                film_name = name
                break
        if film_name is None:
            raise ValueError(f"Hole {hole_name!r} is not contained in any film.")
        return self.polygon_fluxoid(points, film=film_name, interp_method=interp_method,
                                    units=units, with_units=with_units)
