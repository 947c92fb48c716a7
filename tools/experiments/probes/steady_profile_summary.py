#!/usr/bin/env python3
"""Per-kernel totals of the last forward in a tools/steady_profile.py kernel trace (everything after the longest gap)."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
gaps = [(int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp']), i) for i in range(len(rows) - 1)]
_, cut = max(gaps)
last = rows[cut + 1:]
tot = collections.Counter(); cnt = collections.Counter()
for r in last:
    n = r['Kernel_Name'][:70]; d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot[n] += d; cnt[n] += 1
span = int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])
print('last forward: %d kernels, busy %.2f ms, span %.2f ms' % (len(last), sum(tot.values()) / 1e6, span / 1e6))
for n, d in tot.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    print('%-70s calls %4d  total %7.2f ms  avg %8.1f us' % (n, cnt[n], d / 1e6, d / cnt[n] / 1e3))
