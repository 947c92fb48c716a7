#!/usr/bin/env python3
"""Summarise a tools/prof_w43.sh output directory: per F(4x4, 3x3) kernel form and grid size the mean duration and the mean counter values
per dispatch; MFMA-busy share of the SIMD cycles, waiting share of the wave cycles, per-wave-chunk instruction counts."""
import collections, csv, glob, sys
root = sys.argv[1]


def short(name):
    if 'wino43' in name and 'conv3x3_wrw_gen' in name:
        return 'wino43 weight grad'
    if 'wino43' in name and 'conv3x3_gen' in name:
        return 'wino43 generated'
    if 'wino43' in name and 'conv3x3<' in name:
        return 'wino43 r04 ' + ('8 waves' if ', true,' in name.split('(')[0] else '4 waves')
    return None


for p in sorted(glob.glob(root + '/trace/*/*_kernel_trace.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        k = short(r['Kernel_Name'])
        if k:
            agg[(k, int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('== kernel trace: mean / min duration per (kernel, grid size)')
    for (k, g), v in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print('  %-20s grid %8d  launches %4d  mean %8.1f us  min %8.1f us' % (k, g, len(v), sum(v) / len(v), min(v)))
allm = collections.defaultdict(dict)
for p in sorted(glob.glob(root + '/pmc*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = short(r['Kernel_Name'])
        if k:
            agg[(k, int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    for key, d in agg.items():
        allm[key].update({c: sum(v) / len(v) for c, v in d.items()})
print('== counters (mean per dispatch)')
for (k, g), m in sorted(allm.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print('  %-20s grid %8d ' % (k, g) + ' '.join('%s=%.4g' % (c, v) for c, v in sorted(m.items())))
    line = '      '
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and m.get('GRBM_GUI_ACTIVE'):
        line += 'MFMA-busy share of SIMD cycles = %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * m['GRBM_GUI_ACTIVE'] / 8))
    if 'SQ_WAIT_ANY' in m and m.get('SQ_WAVE_CYCLES'):
        line += '  | s_waitcnt / barrier share of wave cycles = %.3f' % (m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'])
    if 'SQ_WAIT_INST_ANY' in m and m.get('SQ_WAVE_CYCLES'):
        line += '  | issue-stall share = %.3f' % (m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'])
    if 'SQ_ACTIVE_INST_ANY' in m and m.get('SQ_WAVE_CYCLES'):
        line += '  | issuing share = %.3f' % (m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'])
    if 'SQ_INSTS_MFMA' in m and m['SQ_INSTS_MFMA']:
        chunks = m['SQ_INSTS_MFMA'] / 36.0            # wave-chunks
        line += '  | per wave and chunk: VALU (non-MFMA) %.1f  LDS %.1f  VMEM %.1f  SALU %.1f' % (
            (m.get('SQ_INSTS_VALU', 0) - m['SQ_INSTS_MFMA']) / chunks, m.get('SQ_INSTS_LDS', 0) / chunks, m.get('SQ_INSTS_VMEM', 0) / chunks,
            m.get('SQ_INSTS_SALU', 0) / chunks)
    print(line)
