#!/usr/bin/env python3
"""Condenses tools/collect_r02_evidence.sh's rocprofv3 output into the files committed under profiles/."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
res_dir = os.path.join(root, 'summaries')
os.makedirs(res_dir, exist_ok=True)


def short(name):
    return name.split('(')[0][:110]


def stats_rows(sub):
    for p in glob.glob('%s/%s/*/*_kernel_stats.csv' % (root, sub)):
        rows = list(csv.DictReader(open(p)))
        rows.sort(key=lambda r: -float(r['TotalDurationNs']))
        return rows
    return []


def trace_rows(sub):
    for p in glob.glob('%s/%s/*/*_kernel_trace.csv' % (root, sub)):
        return list(csv.DictReader(open(p)))
    return []


def counters(sub, match):
    """{kernel: {counter: [values]}} over dispatches whose kernel name contains `match`."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob('%s/%s/*/*_counter_collection.csv' % (root, sub)):
        for r in csv.DictReader(open(p)):
            if match in r['Kernel_Name']:
                agg[(short(r['Kernel_Name']), int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


# ---- bench: kernel stats, sepconv forward by grid
rows = stats_rows('bench_trace')
with open(os.path.join(res_dir, 'r02_bench_kernel_stats_top40.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows[:40]:
        w.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
by_grid = collections.defaultdict(list)
for r in trace_rows('bench_trace'):
    if 'sepconv_forward' in r['Kernel_Name']:
        by_grid[(short(r['Kernel_Name']), int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
lines = ['rocprofv3 --kernel-trace of `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline` (r02), sepconv forward launches by grid size',
         '(grid 131072 threads = 256 workgroups x 512 = the [32,1,128,128] roofline shape; grid 655360 = the in-model call over all 5 time steps, [160,1,128,128])']
fwd_avg = None
for (k, g), v in sorted(by_grid.items()):
    v = sorted(v)
    lines.append('%s grid %d: %d launches, mean %.2f us, median %.2f us, min %.2f us' % (k, g, len(v), sum(v) / len(v), v[len(v) // 2], v[0]))
    if g == 131072:
        fwd_avg = {'launches': len(v), 'mean_us': sum(v) / len(v), 'median_us': v[len(v) // 2], 'min_us': v[0], 'kernel': k}
open(os.path.join(res_dir, 'r02_sepconv_fwd_by_grid.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

# ---- sepconv forward PMC (in-bench launches of the roofline shape)
fetch = counters('bench_fetch', 'sepconv_forward')
write = counters('bench_write', 'sepconv_forward')
pmc = {'shape': [32, 1, 128, 128], 'ks': 51, 'algorithmic_bytes': 220062208}
for (k, g), d in fetch.items():
    if g == 131072:
        pmc['kernel'] = k
        pmc['FETCH_SIZE_KB_mean'] = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE'])
        pmc['n_dispatches'] = len(d['FETCH_SIZE'])
for (k, g), d in write.items():
    if g == 131072:
        pmc['WRITE_SIZE_KB_mean'] = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
if 'FETCH_SIZE_KB_mean' in pmc and 'WRITE_SIZE_KB_mean' in pmc:
    pmc['hbm_bytes_per_launch'] = int(2 * pmc['FETCH_SIZE_KB_mean'] * 1024 + pmc['WRITE_SIZE_KB_mean'] * 1024)
    pmc['traffic_over_algorithmic'] = round(pmc['hbm_bytes_per_launch'] / pmc['algorithmic_bytes'], 4)
pmc['kernel_trace_in_bench'] = fwd_avg
pmc['note'] = ('hbm_bytes = 2 x FETCH_SIZE(KB) x 1024 + WRITE_SIZE(KB) x 1024: MI355X_MICROARCH.md (HBM) -- on gfx950 FETCH_SIZE reports exactly '
               'half the bytes of a wide coalesced streaming read (16 B/lane global_load and LDS-DMA alike); WRITE_SIZE is exact for 16 B/lane '
               'streaming stores. Counters from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE GRBM_GUI_ACTIVE` (separate runs) of '
               '`python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline`, dispatches of grid 131072 only.')
try:       # what bench.py checks before it prints the value as `roofline.traffic`
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from video_frame_inpainting_amd import _native
    pmc['library_version'] = int(_native.lib().tai_sepconv_version())
    pmc['forward_variant'] = int(_native.lib().tai_sepconv_default_forward_variant(1, 128, 51))
except Exception as e:
    pmc['library_version_error'] = repr(e)
pmc['source'] = ('tools/collect_r02_evidence.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE GRBM_GUI_ACTIVE / --kernel-trace --stats, separate '
                 'runs of `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline`; dispatches of the [32,1,128,128] roofline shape (grid 131072)')
json.dump(pmc, open(os.path.join(res_dir, 'sepconv_fwd_pmc_measured.json'), 'w'), indent=1)
print(json.dumps(pmc, indent=1)[:600])

# ---- Winograd kernel counters (all wino::conv3x3 dispatches of the bench run, summed)
wino = {}
for sub in ('bench_sq', 'bench_lds', 'bench_fetch', 'bench_write'):
    for (k, g), d in counters(sub, 'wino').items():
        for c, v in d.items():
            wino.setdefault(c, [0.0, 0])
            wino[c][0] += sum(v)
            wino[c][1] += len(v)
lines = ['rocprofv3 PMC counters of the Winograd convolution kernels (wino::conv3x3<...>) inside `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline`,',
         'summed over every dispatch of the run (eager warm-up forwards + graph replays); separate --pmc passes']
for c, (tot, n) in sorted(wino.items()):
    lines.append('%-28s sum %.6g over %d dispatches (mean %.6g)' % (c, tot, n, tot / max(n, 1)))
if 'SQ_VALU_MFMA_BUSY_CYCLES' in wino and 'SQ_BUSY_CYCLES' in wino:
    lines.append('SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES) = %.3f' % (wino['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (4 * wino['SQ_BUSY_CYCLES'][0])))
if 'SQ_LDS_BANK_CONFLICT' in wino and 'SQ_LDS_IDX_ACTIVE' in wino:
    lines.append('SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f' % (wino['SQ_LDS_BANK_CONFLICT'][0] / wino['SQ_LDS_IDX_ACTIVE'][0]))
wrows = [r for r in rows if 'wino' in r['Name']]
tot = sum(float(r['TotalDurationNs']) for r in wrows)
lines.append('kernel-trace: %d wino dispatches, %.2f ms in total = %.1f %% of the GPU time of the run' % (sum(int(r['Calls']) for r in wrows), tot / 1e6, sum(float(r['Percentage']) for r in wrows)))
open(os.path.join(res_dir, 'r02_wino_conv_pmc.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

# ---- sepconv forward, C = 3 (cfg4 shape)
c3 = {'shape': [16, 3, 256, 256], 'ks': 51, 'algorithmic_bytes': 28648752 * 16, 'factored_flops': 2 * 16 * 3 * 256 * 256 * (51 * 51 + 51)}
for r in stats_rows('c3_trace'):
    if 'sepconv_forward' in r['Name'] and int(r['Calls']) >= 20:
        c3['kernel'] = short(r['Name']); c3['avg_us'] = float(r['AverageNs']) / 1e3; c3['min_us'] = float(r['MinNs']) / 1e3; c3['calls'] = int(r['Calls'])
for sub, key in (('c3_fetch', 'FETCH_SIZE'), ('c3_write', 'WRITE_SIZE')):
    for (k, g), d in counters(sub, 'sepconv_forward').items():
        if key in d and g >= 100000:
            c3[key + '_KB_mean'] = sum(d[key]) / len(d[key])
for (k, g), d in counters('c3_sq', 'sepconv_forward').items():
    if g >= 100000:
        c3['sq'] = {c: sum(v) / len(v) for c, v in d.items()}
if 'FETCH_SIZE_KB_mean' in c3 and 'WRITE_SIZE_KB_mean' in c3:
    c3['hbm_bytes_per_launch'] = int(2 * c3['FETCH_SIZE_KB_mean'] * 1024 + c3['WRITE_SIZE_KB_mean'] * 1024)
    c3['traffic_over_algorithmic'] = round(c3['hbm_bytes_per_launch'] / c3['algorithmic_bytes'], 3)
if 'avg_us' in c3:
    c3['ceilings'] = {'hbm_TBps_algorithmic': round(c3['algorithmic_bytes'] / c3['avg_us'] / 1e6, 3), 'frac_of_8TBps': round(c3['algorithmic_bytes'] / c3['avg_us'] / 1e6 / 8, 3),
                      'TFLOPs_factored': round(c3['factored_flops'] / c3['avg_us'] / 1e6, 1), 'frac_of_157.3TF_fp32_vector': round(c3['factored_flops'] / c3['avg_us'] / 1e6 / 157.3, 3)}
json.dump(c3, open(os.path.join(res_dir, 'sepconv_fwd_pmc_c3.json'), 'w'), indent=1)
print(json.dumps(c3, indent=1))
