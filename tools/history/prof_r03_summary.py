#!/usr/bin/env python3
"""Condenses tools/collect_r03_evidence.sh's rocprofv3 output into the files committed under profiles/."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
res_dir = os.path.join(root, 'summaries')
os.makedirs(res_dir, exist_ok=True)
CMD = 'python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras'


def short(name):
    return name.split('(')[0][:110]


def newest(pattern):
    """Only the newest matching file: gpurun merges a re-run's output into the same directory under a new process id."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def stats_rows(sub):
    for p in newest('%s/%s/*/*_kernel_stats.csv' % (root, sub)):
        rows = list(csv.DictReader(open(p)))
        rows.sort(key=lambda r: -float(r['TotalDurationNs']))
        return rows
    return []


def trace_rows(sub):
    for p in newest('%s/%s/*/*_kernel_trace.csv' % (root, sub)):
        return list(csv.DictReader(open(p)))
    return []


def counters(sub, match):
    """{(kernel, grid): {counter: [values]}}, plus per-dispatch durations under '_us', over dispatches whose name contains `match`."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in newest('%s/%s/*/*_counter_collection.csv' % (root, sub)):
        seen = set()
        for r in csv.DictReader(open(p)):
            if match in r['Kernel_Name']:
                key = (short(r['Kernel_Name']), int(r['Grid_Size']))
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
                if r['Dispatch_Id'] not in seen:
                    seen.add(r['Dispatch_Id'])
                    agg[key]['_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return agg


mean = lambda v: sum(v) / len(v) if v else float('nan')

# ---- bench: kernel stats
rows = stats_rows('bench_trace')
with open(os.path.join(res_dir, 'r03_bench_kernel_stats_top40.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows[:40]:
        w.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])

# ---- sepconv forward by grid: trace durations and counters for BOTH launch shapes
def launch_class(kernel, grid):
    """131072 = the [32,1,128,128] roofline launch; 655360 = the in-model launch [160,1,128,128] -- one-tile kernels have that grid,
    the persistent kernel (one workgroup per CU: grid 131072 as well) is told by its name."""
    if 'persistent' in kernel:
        return 655360
    return grid


by_grid = collections.defaultdict(list)
for r in trace_rows('bench_trace'):
    if 'sepconv_forward' in r['Kernel_Name']:
        k = short(r['Kernel_Name'])
        by_grid[(k, launch_class(k, int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
ALG = {131072: 220062208, 655360: 1100311040}
lines = ['rocprofv3 of `%s` (r03), sepconv forward launches by grid size' % CMD,
         '([32,1,128,128] = the roofline shape, 256 one-tile workgroups, re-read by back-to-back graph replays: Infinity-Cache-warm;',
         ' [160,1,128,128] = the in-model launch over all 5 time steps, its 1.07 GB of taps just written by the preceding convolutions; kernel 20: 256 persistent workgroups)',
         '', '-- kernel trace (--kernel-trace --stats run)']
fwd_avg = {}
for (k, g), v in sorted(by_grid.items()):
    v = sorted(v)
    m = mean(v)
    frac = ALG.get(g, 0) / (m * 1e-6) / 8e12 if g in ALG else float('nan')
    lines.append('%s %s: %d launches, mean %.2f us, median %.2f us, min %.2f us  -> %.3f of 8 TB/s by the mean (%.3f by the median)' % (
        k, {131072: '[32,1,128,128]', 655360: '[160,1,128,128] (in-model)'}.get(g, 'grid %d' % g), len(v), m, v[len(v) // 2], v[0], frac,
        ALG.get(g, 0) / (v[len(v) // 2] * 1e-6) / 8e12 if g in ALG else float('nan')))
    fwd_avg[g] = {'launches': len(v), 'mean_us': m, 'median_us': v[len(v) // 2], 'min_us': v[0], 'kernel': k}
lines += ['', '-- counters (separate --pmc passes of the same command; per-dispatch means)']
per_grid = collections.defaultdict(dict)
for sub in ('bench_fetch', 'bench_write', 'bench_sq', 'bench_lds'):
    for (k, g), d in counters(sub, 'sepconv_forward').items():
        for c, v in d.items():
            per_grid[launch_class(k, g)][(sub, c)] = mean(v)
for g in sorted(per_grid):
    d = per_grid[g]
    fetch, write = d.get(('bench_fetch', 'FETCH_SIZE')), d.get(('bench_write', 'WRITE_SIZE'))
    lines.append('%s:' % {131072: '[32,1,128,128] (grid 131072, fwd::sepconv_forward_ab<5, 0>)', 655360: '[160,1,128,128] in-model (fwd::sepconv_forward_persistent, 256 workgroups)'}.get(g, 'grid %d' % g))
    if fetch is not None and write is not None:
        hbm = 2 * fetch * 1024 + write * 1024
        lines.append('  FETCH_SIZE %.0f KB (x2: gfx950 half-count for 16 B/lane reads) + WRITE_SIZE %.0f KB = %.1f MB per launch = %.4f x the algorithmic %d B' % (
            fetch, write, hbm / 1e6, hbm / ALG.get(g, 1), ALG.get(g, 0)))
    ga, us = d.get(('bench_write', 'GRBM_GUI_ACTIVE')), d.get(('bench_write', '_us'))
    if ga and us:
        lines.append('  GRBM_GUI_ACTIVE %.4g (sum over 8 XCDs) over %.1f us -> %.2f GHz average clock (counter runs are serialised: not the clock of a replayed graph)' % (ga, us, ga / 8 / us / 1e3))
    for c in ('SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_WAIT_ANY'):
        if ('bench_sq', c) in d:
            lines.append('  %-18s %.5g' % (c, d[('bench_sq', c)]))
    if ('bench_sq', 'SQ_WAIT_ANY') in d and ('bench_sq', 'SQ_WAVE_CYCLES') in d:
        lines.append('  SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f;  wave lifetime = 4 x SQ_WAVE_CYCLES / SQ_WAVES = %.0f cycles' % (
            d[('bench_sq', 'SQ_WAIT_ANY')] / d[('bench_sq', 'SQ_WAVE_CYCLES')], 4 * d[('bench_sq', 'SQ_WAVE_CYCLES')] / d[('bench_sq', 'SQ_WAVES')]))
    if ('bench_lds', 'SQ_LDS_BANK_CONFLICT') in d:
        lines.append('  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f;  SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY = %.3f' % (
            d[('bench_lds', 'SQ_LDS_BANK_CONFLICT')] / d[('bench_lds', 'SQ_LDS_IDX_ACTIVE')], d[('bench_lds', 'SQ_ACTIVE_INST_VALU')] / d[('bench_lds', 'SQ_ACTIVE_INST_ANY')]))
open(os.path.join(res_dir, 'r03_sepconv_fwd_by_grid.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

# ---- sepconv forward PMC json (what bench.py reads for roofline.traffic)
pmc = {'shape': [32, 1, 128, 128], 'ks': 51, 'algorithmic_bytes': 220062208}
d = per_grid.get(131072, {})
if ('bench_fetch', 'FETCH_SIZE') in d and ('bench_write', 'WRITE_SIZE') in d:
    pmc['FETCH_SIZE_KB_mean'] = d[('bench_fetch', 'FETCH_SIZE')]
    pmc['WRITE_SIZE_KB_mean'] = d[('bench_write', 'WRITE_SIZE')]
    pmc['hbm_bytes_per_launch'] = int(2 * pmc['FETCH_SIZE_KB_mean'] * 1024 + pmc['WRITE_SIZE_KB_mean'] * 1024)
    pmc['traffic_over_algorithmic'] = round(pmc['hbm_bytes_per_launch'] / pmc['algorithmic_bytes'], 4)
d2 = per_grid.get(655360, {})
if ('bench_fetch', 'FETCH_SIZE') in d2 and ('bench_write', 'WRITE_SIZE') in d2:
    pmc['in_model'] = {'shape': [160, 1, 128, 128], 'algorithmic_bytes': 1100311040,
                       'hbm_bytes_per_launch': int(2 * d2[('bench_fetch', 'FETCH_SIZE')] * 1024 + d2[('bench_write', 'WRITE_SIZE')] * 1024),
                       'kernel_trace_in_bench': fwd_avg.get(655360)}
    pmc['in_model']['traffic_over_algorithmic'] = round(pmc['in_model']['hbm_bytes_per_launch'] / 1100311040, 4)
pmc['kernel_trace_in_bench'] = fwd_avg.get(131072)
pmc['note'] = ('hbm_bytes = 2 x FETCH_SIZE(KB) x 1024 + WRITE_SIZE(KB) x 1024: MI355X_MICROARCH.md (HBM) -- on gfx950 FETCH_SIZE reports exactly '
               'half the bytes of a wide coalesced streaming read (16 B/lane global_load and LDS-DMA alike) and counts Infinity-Cache hits; '
               'WRITE_SIZE is exact for 16 B/lane streaming stores.')
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from video_frame_inpainting_amd import _native
    pmc['library_version'] = int(_native.lib().tai_sepconv_version())
    pmc['forward_variant'] = int(_native.lib().tai_sepconv_default_forward_variant(1, 128, 51))
except Exception as e:
    pmc['library_version_error'] = repr(e)
pmc['source'] = 'tools/collect_r03_evidence.sh: separate rocprofv3 runs (--pmc FETCH_SIZE / --pmc WRITE_SIZE GRBM_GUI_ACTIVE / --kernel-trace --stats) of `%s`' % CMD
json.dump(pmc, open(os.path.join(res_dir, 'sepconv_fwd_pmc.json'), 'w'), indent=1)

# ---- Winograd kernel: per template instance and grid, MFMA-busy share of the SIMD cycles
tr = collections.defaultdict(list)
for r in trace_rows('bench_trace'):
    if 'wino::conv3x3' in r['Kernel_Name']:
        tr[(short(r['Kernel_Name']), int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
sq = counters('bench_sq', 'wino::conv3x3')
lds = counters('bench_lds', 'wino::conv3x3')
lines = ['rocprofv3 of `%s` (r03): the Winograd convolution kernel per template instance and grid' % CMD,
         'MFMA share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the fraction of SIMD-cycles in which the MFMA pipe was busy',
         '(padded channels included: K = 51 runs as 64, C = 51 as 56); us = kernel-trace mean; counters from the serialised --pmc pass',
         '', '%-44s %9s %6s %9s %11s %10s %9s' % ('kernel<ACT,DBG,SKIP,PARTS,TALL,EPI>', 'grid', 'n', 'mean us', 'MFMA share', 'LDS confl', 'wait any')]
tot_busy = tot_cyc = 0.0
for key in sorted(tr, key=lambda k: -sum(tr[k])):
    k, g = key
    s, l = sq.get(key, {}), lds.get(key, {})
    share = mean(s['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024 * mean(s['GRBM_GUI_ACTIVE']) / 8) if s.get('GRBM_GUI_ACTIVE') else float('nan')
    if s.get('GRBM_GUI_ACTIVE'):
        tot_busy += sum(s['SQ_VALU_MFMA_BUSY_CYCLES']); tot_cyc += 1024 * sum(s['GRBM_GUI_ACTIVE']) / 8
    confl = mean(l['SQ_LDS_BANK_CONFLICT']) / mean(l['SQ_LDS_IDX_ACTIVE']) if l.get('SQ_LDS_IDX_ACTIVE') else float('nan')
    wait = mean(s['SQ_WAIT_ANY']) / mean(s['SQ_WAVE_CYCLES']) if s.get('SQ_WAVE_CYCLES') else float('nan')
    lines.append('%-44s %9d %6d %9.1f %11.3f %10.3f %9.3f' % (k.replace('void wino::', '')[:44], g, len(tr[key]), mean(tr[key]), share, confl, wait))
if tot_cyc:
    lines.append('all dispatches: MFMA share %.3f of the SIMD-cycles' % (tot_busy / tot_cyc))
wrows = [r for r in rows if 'wino::conv3x3' in r['Name']]
lines.append('kernel-trace: %d dispatches, %.2f ms in total = %.1f %% of the GPU time of the run' % (
    sum(int(r['Calls']) for r in wrows), sum(float(r['TotalDurationNs']) for r in wrows) / 1e6, sum(float(r['Percentage']) for r in wrows)))
open(os.path.join(res_dir, 'r03_wino_conv_pmc.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

# ---- sepconv forward, C = 3 (cfg4 shape): kernels 17 and 19 from the trace, counters of 19
c3 = {'shape': [16, 3, 256, 256], 'ks': 51, 'algorithmic_bytes': 28648752 * 16, 'factored_flops': 2 * 16 * 3 * 256 * 256 * (51 * 51 + 51)}
for r in stats_rows('c3_trace'):
    if 'sepconv_forward_asm_c3' in r['Name'] and int(r['Calls']) >= 20:
        tag = 'kernel_19_dma_staging' if '<true>' in r['Name'] or 'Lb1' in r['Name'] else 'kernel_17_register_staging'
        c3[tag] = {'kernel': short(r['Name']), 'avg_us': float(r['AverageNs']) / 1e3, 'min_us': float(r['MinNs']) / 1e3, 'calls': int(r['Calls'])}
for sub, key in (('c3_fetch', 'FETCH_SIZE'), ('c3_write', 'WRITE_SIZE')):
    for (k, g), d in counters(sub, 'sepconv_forward').items():
        if key in d and g >= 100000:
            c3[key + '_KB_mean'] = mean(d[key])
for (k, g), d in counters('c3_sq', 'sepconv_forward').items():
    if g >= 100000:
        c3['sq'] = {c: mean(v) for c, v in d.items()}
if 'FETCH_SIZE_KB_mean' in c3 and 'WRITE_SIZE_KB_mean' in c3:
    c3['hbm_bytes_per_launch'] = int(2 * c3['FETCH_SIZE_KB_mean'] * 1024 + c3['WRITE_SIZE_KB_mean'] * 1024)
    c3['traffic_over_algorithmic'] = round(c3['hbm_bytes_per_launch'] / c3['algorithmic_bytes'], 3)
if 'kernel_19_dma_staging' in c3:
    us = c3['kernel_19_dma_staging']['avg_us']
    c3['ceilings'] = {'hbm_TBps_algorithmic': round(c3['algorithmic_bytes'] / us / 1e6, 3), 'frac_of_8TBps': round(c3['algorithmic_bytes'] / us / 1e6 / 8, 3),
                      'TFLOPs_factored': round(c3['factored_flops'] / us / 1e6, 1), 'frac_of_157.3TF_fp32_vector': round(c3['factored_flops'] / us / 1e6 / 157.3, 3)}
try:
    c3['library_version'] = pmc.get('library_version')
except Exception:
    pass
json.dump(c3, open(os.path.join(res_dir, 'sepconv_fwd_pmc_c3.json'), 'w'), indent=1)
print(json.dumps(c3, indent=1))
