"""Builds the C restatement of the reference's numba kernels (test infrastructure).

    python oracle/build_oracle.py      ->  oracle/_build/liboracle_kernels.so

The reference itself is pure Python (no C/C++ sources to compile from /root/reference), so
there is no ``oracle/_ref`` build: "reference unbuildable as native code" is by construction,
see DESIGN.md.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "liboracle_kernels.so")


def build(verbose: bool = True) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = os.path.join(HERE, "csrc", "oracle_kernels.c")
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(src):
        return OUT
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found")
    cmd = [gcc, "-O3", "-ffast-math", "-fopenmp", "-fPIC", "-shared", src, "-o", OUT, "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build())
