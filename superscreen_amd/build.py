"""Builds ``superscreen_amd/lib/libsuperscreen_hip.so`` for gfx950 with hipcc.

One translation unit per ``csrc/*.hip`` file, compiled in parallel, linked into a single
shared library that exports exactly the ``extern "C"`` symbols declared in
``include/superscreen_hip.h``.  hipcc cross-compiles without a GPU, so this runs in the
build container; the resulting ``.so`` is git-ignored but travels to the GPU box.

    python -m superscreen_amd.build [--force]
"""
from __future__ import annotations

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
           "lu.hip", "chol.hip", "collective.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build the HIP library.")
    return exe


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


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

    def compile_one(src: str) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if not force and _newer(o, [s] + headers):
            return o
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    out = os.path.join(LIBDIR, libname)
    if force or not _newer(out, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    names = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--libname=")]
    print(build(force="--force" in sys.argv, extra_flags=flags, libname=names[0] if names else LIBNAME))
