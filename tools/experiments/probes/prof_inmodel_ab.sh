#!/bin/bash
# rocprofv3 kernel trace of tools/sepconv_inmodel_ab.py: the sepconv node's duration inside the replayed bi-TAI forward, two forward
# variants captured side by side on the same box.  Usage (GPU box, repo root): tools/prof_inmodel_ab.sh <tag> <variants, e.g. 23,26>
set -o pipefail
out=gpurun_out/inmodel_ab_$1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/sepconv_inmodel_ab.py $2 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
grep "per replayed forward" $out/run.log
python3 - $out <<'PY'
import csv, glob, sys, collections
rows = []
for p in sorted(glob.glob(sys.argv[1] + '/trace/*/*_kernel_trace.csv')):
    rows = list(csv.DictReader(open(p)))
agg = collections.defaultdict(list)
for r in rows:
    if 'sepconv_forward' in r['Kernel_Name']:
        agg[(r['Kernel_Name'].split('(')[0], r.get('Grid_Size', r.get('Grid_Size_X')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (k, g), v in sorted(agg.items()):
    v2 = sorted(v)
    print('%s grid %s: %d launches, mean %.1f us, median %.1f, min %.1f, max %.1f' % (k, g, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0], v2[-1]))
PY
