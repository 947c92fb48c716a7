#!/usr/bin/env python3
"""Summarise a tools/prof_wrw.sh output directory: per kernel mean duration and mean counter values per dispatch."""
import collections, csv, glob, sys
root = sys.argv[1]
want = ('conv3x3_wrw', 'wrw_reduce', 'igemm_wrw', 'batched_transpose')
for p in sorted(glob.glob(root + '/trace/*/*_kernel_stats.csv')):
    print('== kernel stats')
    for r in csv.DictReader(open(p)):
        if any(w in r['Name'] for w in want):
            print('  %-60s calls=%s avg=%.1f us min=%.1f us' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
for p in sorted(glob.glob(root + '/pmc*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if 'conv3x3_wrw' not in r['Kernel_Name']:
            continue
        agg[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', p.split('/')[-3])
    for k, d in agg.items():
        print('  ', k, {c: round(sum(v) / len(v)) for c, v in d.items()})
