#!/usr/bin/env python3
"""tools/prof_bwd.sh's summary.txt -> profiles/sepconv_bwd_pmc.json (medians, counters and HBM bytes of the default backward kernels).
Usage: python tools/prof_bwd_json.py gpurun_out/prof_bwd_r03/summary.txt <library version>"""
import ast, json, re, sys
txt = open(sys.argv[1]).read()
out = {'shape': [32, 1, 128, 128], 'ks': 51, 'algorithmic_bytes_all_three_gradients': 438027264, 'library_version': int(sys.argv[2]), 'kernels': {}}
tr, cnt = {}, {}
for l in txt.split('== kernel trace, big dispatches only')[1].split('== pmc1')[0].strip().splitlines():
    m = re.match(r'\s*(?:void )?(bwd::.*?\S)\s+n=(\d+) median=([\d.]+) us min=([\d.]+) us', l)
    if m:
        tr[m.group(1)] = {'n': int(m.group(2)), 'median_us': float(m.group(3)), 'min_us': float(m.group(4))}
for blk in txt.split('== pmc')[1:]:
    for l in blk.splitlines()[1:]:
        m = re.match(r'\s*(?:void )?(bwd::.*?\S) (\{.*\})', l)
        if m:
            cnt.setdefault(m.group(1), {}).update(ast.literal_eval(m.group(2)))
DEFAULT = ('bwd::sepconv_grad_i_strips_asm<1>', 'bwd::sepconv_grad_i_reduce', 'bwd::sepconv_grad_vh_ab<true, 1>')
for k in DEFAULT + ('bwd::sepconv_grad_vh_ab<true, 0>', 'bwd::sepconv_grad_vh_ab<true, 2>', 'bwd::sepconv_grad_vh_ab<false, 0>'):
    d, c = dict(tr.get(k, {})), cnt.get(k, {})
    for key in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'GRBM_GUI_ACTIVE'):
        if key in c:
            d[key + ('_KB' if key.endswith('SIZE') else '')] = c[key]
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        d['hbm_bytes_per_launch'] = int(2 * c['FETCH_SIZE'] * 1024 + c['WRITE_SIZE'] * 1024)
    out['kernels'][k] = d
ks = out['kernels']
tot = sum(ks[k]['median_us'] for k in DEFAULT)
out['all_three_gradients_us_sum_of_medians'] = round(tot, 1)
out['frac_of_8TBps'] = round(438027264 / (tot * 1e-6) / 8e12, 4)
hb = sum(ks[k].get('hbm_bytes_per_launch', 0) for k in DEFAULT)
out['hbm_bytes_all_three'] = hb
out['traffic_over_algorithmic'] = round(hb / 438027264, 3)
out['source'] = ('tools/prof_bwd.sh (rocprofv3 kernel trace + separate --pmc passes of tools/sepconv_bwd_bench.py 32 1 128 128); '
                 'vh_ab<true, 1> = default (tap loads at entry, gV waves at priority 1), <true, 0> / <true, 2> = gV waves at 0 (round 3) / 2, <false, 0> = round 2')
json.dump(out, open('profiles/sepconv_bwd_pmc.json', 'w'), indent=1)
print(out['all_three_gradients_us_sum_of_medians'], out['frac_of_8TBps'], out['traffic_over_algorithmic'])
