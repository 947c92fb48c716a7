#!/usr/bin/env python3
"""Summarise a tools/prof_fwd.sh output directory: per kernel (big dispatches only) mean counter values."""
import collections, csv, glob, sys
root = sys.argv[1]
for p in sorted(glob.glob(root + '/trace/*/*_kernel_stats.csv')):
    print('== kernel stats (all dispatches)')
    for r in csv.DictReader(open(p)):
        if 'sepconv' in r['Name']:
            print('  %-70s calls=%s avg=%.1f us min=%.1f us' % (r['Name'].split('(')[0][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
for p in sorted(glob.glob(root + '/trace/*/*_kernel_trace.csv')):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        gs = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
        if 'sepconv' in r['Kernel_Name'] and gs >= 100000:
            d[r['Kernel_Name'].split('(')[0][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('== kernel trace, big dispatches only')
    for k, v in d.items():
        v = sorted(v)
        print('  %-70s n=%d median=%.1f us min=%.1f us' % (k, len(v), v[len(v) // 2], v[0]))
for p in sorted(glob.glob(root + '/pmc*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if 'sepconv' not in r['Kernel_Name'] or int(r['Grid_Size']) < 100000:
            continue
        agg[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', p.split('/')[-3])
    for k, d in agg.items():
        print('  ', k, {c: round(sum(v) / len(v)) for c, v in d.items()})
