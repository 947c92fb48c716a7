#!/usr/bin/env python3
"""Per kernel of a training update: dispatches, total and mean duration, MFMA-busy share of the SIMD cycles and waiting share of the
wave cycles, from ONE rocprofv3 --pmc run (SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY
GRBM_GUI_ACTIVE; durations from the counter file's own timestamps -- counter runs are serialised, so these are not the eager step's
times).  Usage: python tools/train_pmc_summary.py <dir with *_counter_collection.csv> [top N]"""
import collections, csv, glob, sys
root, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = sorted(glob.glob(root + '/*/*_counter_collection.csv'))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
for r in csv.DictReader(open(files[-1])):
    k = r['Kernel_Name'].split('(')[0][:86] + ' g%s' % r['Grid_Size']
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id'], k) not in seen:
        seen.add((r['Dispatch_Id'], k))
        agg[k]['_n'] += 1
        agg[k]['_us'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(d['_us'] for d in agg.values())
print('%-100s %5s %9s %8s %6s %6s' % ('kernel + grid', 'n', 'total ms', 'mean us', 'MFMA', 'wait'))
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['_us'])[:top]:
    mf = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d['GRBM_GUI_ACTIVE'] / 8) if d.get('GRBM_GUI_ACTIVE') else float('nan')
    wt = d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES'] if d.get('SQ_WAVE_CYCLES') else float('nan')
    print('%-100s %5d %9.2f %8.1f %6.3f %6.3f' % (k, d['_n'], d['_us'] / 1e3, d['_us'] / d['_n'], mf, wt))
print('all kernels: %.1f ms' % (tot / 1e3))
