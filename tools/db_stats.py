"""Per-kernel totals from a rocprofv3 (rocpd) results database: python tools/db_stats.py results.db [top]."""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
agg = collections.defaultdict(lambda: [0, 0])
for name, start, end in con.execute("select name, start, end from kernels"):
    agg[name][0] += 1
    agg[name][1] += end - start
total = sum(t for _, t in agg.values())
print(f"total kernel time {total / 1e6:.2f} ms")
for name, (calls, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / 1e6:9.2f} ms {100 * t / total:5.1f}% {calls:6d} calls {t / calls / 1e3:9.1f} us  {name[:120]}")
