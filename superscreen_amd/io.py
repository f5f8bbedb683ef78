"""Persistence for ``Device`` / ``Solution`` / ``FactorizedModel`` (``superscreen/io.py``,
``device/device.py:936-1016``, ``solution.py:936-1087``, ``solver/solve.py:102-180``).

The reference writes HDF5 through ``h5py``.  Every ``to_hdf5`` / ``from_hdf5`` in this package uses only
the small part of the ``h5py.Group`` interface the reference's own methods use (``attrs``,
``create_group``, item assignment of arrays, ``SoftLink``, ``items`` ...), so the same methods run on

* a real ``h5py.File`` / ``h5py.Group`` when ``h5py`` is installed -- same group / dataset / attribute
  names as the reference's files -- and otherwise on
* :class:`File` below: the same tree held in memory and stored as ONE ``numpy`` ``.npz`` archive
  (datasets under their group paths, attributes and links in a JSON member).

``open_file(path, mode)`` picks the backend; ``mode`` is ``"x"`` (create, fail if the file exists),
``"r"`` or ``"r+"`` like the reference's calls.
"""
from __future__ import annotations

import base64
import json
import os
import zipfile
from typing import Any, Dict, Iterator, Optional, Union

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import h5py  # type: ignore
except ImportError:  # the container format below is used instead
    h5py = None

_ATTRS_MEMBER = "__attrs__.json"


class SoftLink:
    """``h5py.SoftLink`` stand-in: an alias for an absolute path in the same file."""

    def __init__(self, path: str):
        self.path = str(path)


class Group:
    """In-memory group tree with the subset of the ``h5py.Group`` interface used by this package."""

    def __init__(self, name: str = "/", parent: Optional["Group"] = None):
        self.name = name
        self.attrs: Dict[str, Any] = {}
        self._items: Dict[str, Union["Group", np.ndarray, SoftLink]] = {}
        self._parent = parent

    # -- tree -----------------------------------------------------------------------------
    @property
    def file(self) -> "Group":
        node = self
        while node._parent is not None:
            node = node._parent
        return node

    def create_group(self, name: str) -> "Group":
        node = self
        for part in [p for p in name.split("/") if p]:
            child = node._items.get(part)
            if child is None:
                child = Group((node.name.rstrip("/") + "/" + part), node)
                node._items[part] = child
            elif not isinstance(child, Group):
                raise ValueError(f"Unable to create group (name already exists: {part!r}).")
            node = child
        if node is self:
            raise ValueError("Empty group name.")
        return node

    def _resolve(self, value):
        seen = 0
        while isinstance(value, SoftLink):
            value = self.file[value.path]
            seen += 1
            if seen > 16:
                raise KeyError("Too many levels of soft links.")
        return value

    def __getitem__(self, path: str):
        node: Any = self.file if path.startswith("/") else self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._items:
                raise KeyError(f"Unable to open object (object {path!r} doesn't exist)")
            node = self._resolve(node._items[part])
        return node

    def __setitem__(self, name: str, value) -> None:
        parts = [p for p in name.split("/") if p]
        node = self.create_group("/".join(parts[:-1])) if len(parts) > 1 else self
        if parts[-1] in node._items:
            raise ValueError(f"Unable to create dataset (name already exists: {parts[-1]!r}).")
        if isinstance(value, (SoftLink, Group)):
            node._items[parts[-1]] = SoftLink(value.name) if isinstance(value, Group) else value
        else:
            node._items[parts[-1]] = np.asarray(value)

    def __contains__(self, path: str) -> bool:
        try:
            self[path]
            return True
        except KeyError:
            return False

    def get(self, path: str, default=None):
        try:
            return self[path]
        except KeyError:
            return default

    def keys(self):
        return self._items.keys()

    def values(self):
        return [self._resolve(v) for v in self._items.values()]

    def items(self):
        return [(k, self._resolve(v)) for k, v in self._items.items()]

    def __iter__(self) -> Iterator[str]:
        return iter(self._items)

    def __len__(self) -> int:
        return len(self._items)

    # -- flattening ------------------------------------------------------------------------
    def _walk(self, arrays: Dict[str, np.ndarray], meta: Dict[str, Any]) -> None:
        entry: Dict[str, Any] = {}
        if self.attrs:
            entry["attrs"] = {k: _encode_attr(v) for k, v in self.attrs.items()}
        links = {k: v.path for k, v in self._items.items() if isinstance(v, SoftLink)}
        if links:
            entry["links"] = links
        meta[self.name] = entry
        for key, value in self._items.items():
            path = self.name.rstrip("/") + "/" + key
            if isinstance(value, Group):
                value._walk(arrays, meta)
            elif isinstance(value, np.ndarray):
                arrays[path.lstrip("/")] = value


def _encode_attr(value):
    if isinstance(value, (bytes, np.void)):
        return {"__bytes__": base64.b64encode(bytes(value)).decode("ascii")}
    if isinstance(value, np.ndarray):
        return {"__ndarray__": value.tolist(), "dtype": str(value.dtype)}
    if isinstance(value, np.generic):
        return value.item()
    if isinstance(value, (str, int, float, bool)) or value is None:
        return value
    raise TypeError(f"Cannot store an attribute of type {type(value).__name__}.")  # like h5py


def _decode_attr(value):
    if isinstance(value, dict) and "__bytes__" in value:
        return np.void(base64.b64decode(value["__bytes__"]))
    if isinstance(value, dict) and "__ndarray__" in value:
        return np.asarray(value["__ndarray__"], dtype=value["dtype"])
    return value


class File(Group):
    """The root group, bound to an ``.npz`` archive.  Context manager; written on close."""

    def __init__(self, path, mode: str = "r"):
        super().__init__("/")
        self.path = os.fspath(path)
        self.mode = mode
        if mode not in ("r", "r+", "x", "w", "a"):
            raise ValueError(f"Invalid mode {mode!r}.")
        exists = os.path.exists(self.path)
        if mode == "x" and exists:
            raise FileExistsError(f"Unable to create file (file exists): {self.path!r}")
        if mode in ("r", "r+") and not exists:
            raise FileNotFoundError(f"Unable to open file (no such file): {self.path!r}")
        if exists and mode in ("r", "r+", "a"):
            self._load()

    def _load(self) -> None:
        with zipfile.ZipFile(self.path) as zf:
            meta = json.loads(zf.read(_ATTRS_MEMBER).decode("utf-8"))
        for name in sorted(meta):  # parents before children
            node = self if name == "/" else self.create_group(name)
            node.attrs.update({k: _decode_attr(v) for k, v in meta[name].get("attrs", {}).items()})
            for key, target in meta[name].get("links", {}).items():
                node._items[key] = SoftLink(target)
        with np.load(self.path, allow_pickle=False) as data:
            for key in data.files:
                if key != _ATTRS_MEMBER:
                    self[key] = data[key]

    def flush(self) -> None:
        if self.mode == "r":
            return
        arrays: Dict[str, np.ndarray] = {}
        meta: Dict[str, Any] = {}
        self._walk(arrays, meta)
        tmp = self.path + ".tmp"
        with zipfile.ZipFile(tmp, "w", compression=zipfile.ZIP_DEFLATED) as zf:
            zf.writestr(_ATTRS_MEMBER, json.dumps(meta))
            for key, value in arrays.items():
                with zf.open(key + ".npy", "w", force_zip64=True) as fh:
                    np.lib.format.write_array(fh, np.ascontiguousarray(value), allow_pickle=False)
        os.replace(tmp, self.path)

    def close(self) -> None:
        self.flush()

    def __enter__(self) -> "File":
        return self

    def __exit__(self, exc_type, exc, tb) -> None:
        if exc_type is None:
            self.close()


def is_group(obj) -> bool:
    return isinstance(obj, Group) or (h5py is not None and isinstance(obj, h5py.Group))


def open_file(path, mode: str = "r"):
    """``h5py.File(path, mode)`` when h5py is installed, otherwise the ``.npz`` container."""
    if h5py is not None:
        return h5py.File(path, mode)
    return File(path, mode)


def soft_link(group, path: str):
    if h5py is not None and isinstance(group, h5py.Group):
        return h5py.SoftLink(path)
    return SoftLink(path)


def serialize_obj(group, obj: Any, name: str, attr: bool = False) -> None:
    """``io.py:8-23``: plain attribute if possible, else a dill pickle under ``<name>.pickle``."""
    import dill

    if attr:
        try:
            if callable(obj) or not isinstance(obj, (str, int, float, bool, np.generic, np.ndarray)):
                raise TypeError
            group.attrs[name] = obj
        except TypeError:
            group.attrs[f"{name}.pickle"] = np.void(dill.dumps(obj))
    else:
        group[f"{name}.pickle"] = np.void(dill.dumps(obj))


def deserialize_obj(group, name: str, attr: bool = False) -> Any:
    """``io.py:26-44``."""
    import dill

    if attr:
        if name in group.attrs:
            return group.attrs[name]
        if f"{name}.pickle" in group.attrs:
            return dill.loads(np.void(group.attrs[f"{name}.pickle"]).tobytes())
    elif f"{name}.pickle" in group:
        return dill.loads(np.void(np.asarray(group[f"{name}.pickle"])[()]).tobytes())
    raise IOError(f"Unable to load {name}.")
