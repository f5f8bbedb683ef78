"""Per-queue summary of one factorization inside a rocprofv3 kernel trace (development aid).
usage: python tools/trace_queues.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
qa = [i for i, r in enumerate(rows) if 'q_assemble' in r['Kernel_Name']]
seg = rows[qa[-2]:]
endi = next(i for i, r in enumerate(seg) if 'film_rhs' in r['Kernel_Name'])
seg = seg[:endi]
t0 = seg[0]['s']
print('factorize span ms', (seg[-1]['e'] - t0) / 1e6)
byq = collections.defaultdict(list)
for r in seg:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = sum(r['e'] - r['s'] for r in rs)
    names = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r['Kernel_Name'][:72]
        names[k][0] += r['e'] - r['s']
        names[k][1] += 1
    print('queue', q, 'n', len(rs), 'busy ms %.1f' % (busy / 1e6), 'first %.2f' % ((rs[0]['s'] - t0) / 1e6),
          'last %.2f' % ((rs[-1]['e'] - t0) / 1e6))
    for k, v in sorted(names.items(), key=lambda kv: -kv[1][0])[:6]:
        print('     %-72s %8.2f ms %5d  avg %.1f us' % (k, v[0] / 1e6, v[1], v[0] / v[1] / 1e3))
