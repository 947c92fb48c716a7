#!/usr/bin/env python3
"""Summarise a tools/prof_split.sh output directory: per Winograd forward kernel (fp32: wino::conv3x3, split bf16: wino::split::conv3x3)
and grid size the mean duration and the mean counter values per dispatch, plus MFMA-busy share of the SIMD cycles and waiting share of
the wave cycles."""
import collections, csv, glob, sys
root = sys.argv[1]


def short(name):
    if 'split' in name and 'conv3x3' in name:
        return 'split::conv3x3'
    if 'wino' in name and 'conv3x3<' in name and 'wrw' not in name:
        return 'wino::conv3x3' + (' (128x32)' if ', true,' in name.split('(')[0] else ' (64x64)')
    return None


for p in sorted(glob.glob(root + '/trace/*/*_kernel_trace.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        k = short(r['Kernel_Name'])
        if k:
            agg[(k, int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('== kernel trace: mean / min duration per (kernel, grid size)')
    for (k, g), v in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print('  %-26s grid %8d  launches %4d  mean %8.1f us  min %8.1f us' % (k, g, len(v), sum(v) / len(v), min(v)))
for p in sorted(glob.glob(root + '/pmc*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = short(r['Kernel_Name'])
        if k:
            agg[(k, int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', p.split('/')[-3])
    for (k, g), d in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        line = '  %-26s grid %8d ' % (k, g) + ' '.join('%s=%.4g' % (c, v) for c, v in sorted(m.items()))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and m.get('GRBM_GUI_ACTIVE'):
            # (as tools/prof_r04_summary.py: the fraction of SIMD-cycles in which the matrix pipe was busy)
            line += '  | MFMA-busy share of SIMD cycles = %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * m['GRBM_GUI_ACTIVE'] / 8))
        if 'SQ_WAIT_ANY' in m and 'SQ_WAVE_CYCLES' in m and m['SQ_WAVE_CYCLES']:
            line += '  | waiting share of wave cycles = %.3f' % (m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'])
        if 'SQ_LDS_BANK_CONFLICT' in m and 'SQ_LDS_IDX_ACTIVE' in m and m['SQ_LDS_IDX_ACTIVE']:
            line += '  | LDS conflict share = %.3f' % (m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'])
        print(line)
