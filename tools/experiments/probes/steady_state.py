#!/usr/bin/env python3
"""Steady-state GPU-time split of bench.py from a rocprofv3 kernel trace: the window covering the last timed steps before the
roofline loop (the 50+ consecutive sepconv launches)."""
import collections, csv, glob, sys
p = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4
ms_per_step = float(sys.argv[3]) if len(sys.argv) > 3 else 140
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = next(i for i in range(len(rows) - 50) if all('sepconv_forward' in names[j] for j in range(i, i + 50)))
tend = int(rows[idx]['Start_Timestamp'])
win = [r for r in rows[:idx] if int(r['Start_Timestamp']) > tend - int(steps * ms_per_step * 1e6)]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    k = r['Kernel_Name'][:100]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in agg.values())
span = (tend - int(win[0]['Start_Timestamp'])) / 1e6
print('window: %d dispatches, GPU busy %.1f ms over %.1f ms (%.0f%%)' % (len(win), tot, span, 100 * tot / span))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%8.2f ms %5.1f%% n=%5d  %s' % (v[1], 100 * v[1] / tot, v[0], k))
