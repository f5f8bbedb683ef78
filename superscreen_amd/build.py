"""Builds ``superscreen_amd/lib/libsuperscreen_hip.so`` for gfx950 with hipcc.

One translation unit per ``csrc/*.hip`` file, compiled in parallel, linked into a single
shared library that exports exactly the ``extern "C"`` symbols declared in
``include/superscreen_hip.h``.  hipcc cross-compiles without a GPU, so this runs in the
build container; the resulting ``.so`` is git-ignored but travels to the GPU box.

    python -m superscreen_amd.build [--force]
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
LIBNAME = "libsuperscreen_hip.so"
ARCH = "gfx950"
SOURCES = ["capi.hip", "assemble.hip", "pairwise.hip", "pairwise_multi.hip", "blas1.hip", "gemm.hip", "gemm_ops.hip",
           "lu.hip", "chol.hip", "chain_streams.hip", "collective.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build the HIP library.")
    return exe


def _digest(paths, extra=()) -> str:
    """sha256 over the CONTENTS of ``paths`` (and the strings of ``extra``): what decides whether an
    object or the library is up to date.  Modification times are not consulted -- a prebuilt file that is
    newer than edited sources (a checkout, a copied tree) must not be reused."""
    h = hashlib.sha256()
    for item in extra:
        h.update(str(item).encode())
        h.update(b"\0")
    for path in paths:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_matches(target: str, digest: str) -> bool:
    try:
        with open(target + ".sha256") as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def _write_stamp(target: str, digest: str) -> None:
    with open(target + ".sha256", "w") as f:
        f.write(digest + "\n")


def build(force: bool = False, verbose: bool = True, extra_flags=(), libname: str = LIBNAME) -> str:
    """``extra_flags`` / ``libname``: experiment builds (e.g. ``-DSSA_SNB=2048`` into a second library that
    ``SSA_LIB_PATH`` selects at run time); they get their own object directory."""
    global OBJDIR
    os.makedirs(LIBDIR, exist_ok=True)
    if libname != LIBNAME:
        OBJDIR = os.path.join(PKG, "build", libname.replace(".so", ""))
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "superscreen_hip.h"))

    flags = FLAGS + list(extra_flags)

    def compile_one(src: str) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        digest = _digest([s] + headers, extra=[hipcc] + flags)
        if not force and _stamp_matches(o, digest):
            return o
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        _write_stamp(o, digest)
        return o

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    out = os.path.join(LIBDIR, libname)
    # the library's stamp is the digest of every source, header and flag it was built from: it travels with the
    # .so (lib/ is not gpurun-ignored), so a box without hipcc can still tell a current library from a stale one
    lib_digest = source_digest(extra_flags)
    if force or not _stamp_matches(out, lib_digest):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        _write_stamp(out, lib_digest)
    return out


def source_digest(extra_flags=()) -> str:
    """Digest of everything the library is built from (sources, headers, flags)."""
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "superscreen_hip.h"))
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + headers, extra=FLAGS + list(extra_flags))


def is_current(libname: str = LIBNAME, extra_flags=()) -> bool:
    """True if ``lib/<libname>`` carries the stamp of the sources in this tree."""
    return _stamp_matches(os.path.join(LIBDIR, libname), source_digest(extra_flags))


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    names = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--libname=")]
    print(build(force="--force" in sys.argv, extra_flags=flags, libname=names[0] if names else LIBNAME))
