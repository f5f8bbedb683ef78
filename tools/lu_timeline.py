"""Chain timeline of the last LU-route factorization inside a rocprofv3 kernel trace (development aid).
usage: python tools/lu_timeline.py <kernel_trace.csv>"""
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
t0 = seg[0]['s']
print('span ms', (seg[-1]['e'] - t0) / 1e6, 'kernels', len(seg))
byq = collections.defaultdict(list)
for r in seg:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = sum(r['e'] - r['s'] for r in rs)
    print('queue', q, 'n', len(rs), 'busy %.1f ms' % (busy / 1e6),
          'first %.2f last %.2f' % ((rs[0]['s'] - t0) / 1e6, (rs[-1]['e'] - t0) / 1e6))
    names = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r['Kernel_Name'][:60]
        names[k][0] += r['e'] - r['s']
        names[k][1] += 1
    for k, v in sorted(names.items(), key=lambda kv: -kv[1][0])[:4]:
        print('     %-60s %7.2f ms  n=%d' % (k, v[0] / 1e6, v[1]))
for q, rs in byq.items():
    d = [r for r in rs if 'diag256' in r['Kernel_Name']]
    if not d:
        continue
    print('queue', q, 'diag launches', len(d))
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    for i in range(0, len(d) - 1, step):
        rnd = (d[i + 1]['s'] - d[i]['s']) / 1e3
        ks = [r for r in rs if d[i]['s'] <= r['s'] < d[i + 1]['s']]
        busy = sum(r['e'] - r['s'] for r in ks) / 1e3
        print('   panel %2d at %6.1f ms: round %7.1f us, chain busy %7.1f us (diag %6.1f), idle %7.1f' %
              (i, (d[i]['s'] - t0) / 1e6, rnd, busy, (d[i]['e'] - d[i]['s']) / 1e3, rnd - busy))
    print('   last diag ends at %.1f ms' % ((d[-1]['e'] - t0) / 1e6))
