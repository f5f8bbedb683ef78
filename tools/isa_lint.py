"""Static check of the built library's gfx950 code: no workgroup barrier with LDS operations still in flight.

``s_barrier`` only synchronises the program counters of a workgroup's waves; an LDS read that has been ISSUED but
not completed when its wave arrives at the barrier is not covered by it.  With hand-rolled barriers
(``__builtin_amdgcn_s_barrier()``; ``__syncthreads()`` carries its own waits) the compiler is free to place the
barrier in front of the ``s_waitcnt lgkmcnt(0)`` of reads whose results are used after it -- which is how a slot of an
LDS ring got refilled (by another wave's DMA) under a read that was still queued (round 5: one factorization in a
hundred of the four-film stack differed in the 10th digit).  Rule checked here, per kernel, in program order:

    every ``s_barrier`` is preceded by an ``s_waitcnt`` with ``lgkmcnt(0)`` that has no LDS instruction
    (``ds_*``) after it.

A HEURISTIC, not a proof: the disassembly is read in fall-through order and only ``ds_*`` operations are tracked;
branches are not followed, so an LDS read that reaches a barrier along another path than the textual one is not seen
(the kernels of this library keep their LDS rings in straight-line loop bodies, which is the shape it does see: it
flags the round-4 form of ``tile_small_nt`` and three siblings).  Scalar memory loads share the counter, so a wait
that covers them covers the LDS operations too.

    python tools/isa_lint.py [path/to/libsuperscreen_hip.so]      exit status 1 if a barrier violates the rule
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _llvm_bin() -> str:
    """The ROCm LLVM tools: $ROCM_PATH/lib/llvm/bin, /opt/rocm/lib/llvm/bin, or wherever llvm-objdump is on PATH."""
    import shutil

    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root and os.path.exists(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump")):
            return os.path.join(root, "lib", "llvm", "bin")
    found = shutil.which("llvm-objdump")
    return os.path.dirname(found) if found else "/opt/rocm/lib/llvm/bin"


LLVM = _llvm_bin()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path: str, workdir: str):
    """The gfx950 code objects embedded in a HIP shared library (one offload bundle per translation unit)."""
    fat = os.path.join(workdir, "fatbin.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for k, b in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(blob)
        bundle = os.path.join(workdir, f"bundle{k}.bin")
        with open(bundle, "wb") as f:
            f.write(blob[b:e])
        co = os.path.join(workdir, f"code{k}.co")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={bundle}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if os.path.getsize(co) > 0:
            out.append(co)
    return out


def lint_disassembly(text: str):
    """Yields (kernel, barrier index, instructions since the last full LDS wait) for every violating barrier."""
    kernel, pending, since, nbar = None, None, [], 0
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            kernel, pending, since, nbar = m.group(1), None, [], 0
            continue
        parts = line.split("//")[0].split()
        if not parts:
            continue
        op = parts[0]
        if op.startswith("ds_"):
            pending = op
            since = [op]
        elif op == "s_waitcnt":
            rest = " ".join(parts[1:])
            # (a bare immediate or a form without lgkmcnt leaves the LDS counter alone)
            if re.search(r"lgkmcnt\(0\)", rest):
                pending, since = None, []
        elif op == "s_barrier":
            nbar += 1
            if pending is not None:
                yield kernel, nbar, list(since)
        if pending is not None and op != pending:
            since.append(op)
            since[:] = since[-6:]


def lint_library(lib_path: str):
    problems, kernels, barriers = [], 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib_path, tmp):
            text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True,
                                  capture_output=True, text=True).stdout
            kernels += len(re.findall(r"^[0-9a-f]+ <.+>:$", text, flags=re.M))
            barriers += len(re.findall(r"\bs_barrier\b", text))
            problems += list(lint_disassembly(text))
    return problems, kernels, barriers


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "superscreen_amd", "lib", "libsuperscreen_hip.so")
    problems, kernels, barriers = lint_library(lib)
    for kernel, k, since in problems:
        try:
            demangled = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip() or kernel
        except OSError:
            demangled = kernel
        print(f"LDS operation in flight at barrier #{k} of {demangled[:160]}: ... {' '.join(since)} s_barrier")
    print(f"{len(problems)} barrier(s) with LDS operations in flight; {barriers} barriers in {kernels} functions checked")
    sys.exit(1 if problems else 0)
