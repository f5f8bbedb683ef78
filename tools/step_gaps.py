"""Where the GPU idles during cold config-H steps (development aid): the gaps of the kernel timeline.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/sg -- python3 tools/step_gaps.py run
    python tools/step_gaps.py analyse /tmp/sg"""
import csv, glob, os, sys


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import superscreen_amd as sc
    from superscreen_amd import synthetic
    device = synthetic.make_stack_device(91, ("washer", "disk"), solve_dtype="float64")
    for i in range(4):
        sc.solve(device=device, applied_field=sc.ConstantField(0.3 + i), iterations=10, progress_bar=False)
    torch.cuda.synchronize()


def analyse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) + glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Direction") or "copy"))
    rows.sort()
    asm = [i for i, r in enumerate(rows) if "system_assemble" in r[2]]
    # steps start at every second system_assemble (two films per step)
    starts = asm[0::2]
    print(f"{len(starts)} steps")
    for si in range(1, len(starts)):
        lo = starts[si]
        hi = starts[si + 1] if si + 1 < len(starts) else len(rows)
        seg = rows[lo:hi]
        # extend back to the first op after the previous step's end: everything between is host time
        prev_end = max(e for s, e, _ in rows[:lo])
        first = seg[0][0]
        end = max(e for s, e, _ in seg)
        # merge busy intervals
        busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
        gaps = []
        for s, e, n in seg[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                if s - cur_e > 20000:
                    gaps.append((s - cur_e, (cur_e - first) / 1e6, n[:50]))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        print(f"step {si}: idle before its first kernel {(first - prev_end) / 1e6:.2f} ms, span {(end - first) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms")
        gaps.sort(reverse=True)
        print("   largest gaps (us @ ms into the step, next op): " + "; ".join(f"{g / 1e3:.0f} @ {t:.1f} {n}" for g, t, n in gaps[:8]))
        print(f"   gaps > 20 us: {len(gaps)}, total {sum(g for g, _, _ in gaps) / 1e6:.2f} ms")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else analyse(sys.argv[2])
