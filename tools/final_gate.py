"""CPU side of the final gate: does a gate log (tools/final_gate.sh, run on the GPU box) belong to THIS tree?

    python tools/final_gate.py --check profiles/r05_final_gate.txt

Recomputes the digest of everything the GPU run depended on (package, kernels, tests, oracle, headers, bench.py,
__graft_entry__.py) and compares it with the `tree_digest` line of the log; also requires the three return codes to be 0.
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tree_digest() -> str:
    h = hashlib.sha256()
    os.chdir(ROOT)
    for root in ("superscreen_amd", "tests", "oracle", "include"):
        for d, _, files in sorted(os.walk(root)):
            if "__pycache__" in d or "/build" in d or d.endswith("/lib") or "_ref" in d:
                continue
            for f in sorted(files):
                if f.endswith((".py", ".hip", ".hpp", ".h", ".c", ".npz")):
                    h.update(os.path.join(d, f).encode())
                    h.update(open(os.path.join(d, f), "rb").read())
    for f in ("bench.py", "__graft_entry__.py"):
        h.update(f.encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--check":
        fields = dict(line.split(None, 1) for line in open(sys.argv[2]).read().splitlines() if " " in line)
        ok = fields.get("tree_digest", "").strip() == tree_digest()
        rcs = [fields.get(k, "?").strip() for k in ("pytest_gpu_rc", "smoke_rc", "bench_rc")]
        print(f"tree digest {'matches' if ok else 'DOES NOT MATCH'} the log; return codes {rcs}")
        sys.exit(0 if ok and rcs == ["0", "0", "0"] else 1)
    print(tree_digest())
