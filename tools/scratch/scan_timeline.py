"""Per-scan device timeline of an odometry run from a rocprofv3 kernel trace: busy / idle time and the kernels of one scan."""
import csv, sys, collections
f = sys.argv[1]; show = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:44]
idx = [i for i, r in enumerate(rows) if nm(r).startswith('k_cov_finalize')]
# steady state: scans 50..end
per = collections.defaultdict(float); n = 0; idle = 0.0; span = 0.0
for a, b in zip(idx[50:-1], idx[51:]):
    seg = rows[a + 1:b + 1]
    s0 = int(rows[a]['End_Timestamp']); e1 = int(rows[b]['End_Timestamp'])
    span += e1 - s0; n += 1
    cur = s0
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > cur: idle += s - cur
        cur = max(cur, e)
        per[nm(r)] += e - s
print('scans', n, 'span/scan us', span / n / 1e3, 'idle/scan us', idle / n / 1e3)
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:25]:
    print('%-46s %8.2f us/scan' % (k, v / n / 1e3))
a, b = idx[show], idx[show + 1]
t0 = int(rows[a]['End_Timestamp']); prev = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f +%6.1f %6.1f us q%s %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), nm(r)))
    prev = max(prev, e)
