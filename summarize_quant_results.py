#!/usr/bin/env python3
"""results.npz files -> ``<metric>_perf_summary.txt`` tables of mean +- standard error over videos, in the reference's
layout (summarize_quant_results.py:217-237; compare quant_tables_orig/*.txt): per-video score = mean over the middle
frames, Mean = mean over videos, StdErr = std / sqrt(N).  Usage:

  python summarize_quant_results.py DEST --results ROOT_A:LabelA ROOT_B:LabelB [--mean_precision 2 --std_err_precision 3]
"""
import argparse
import os

import numpy as np


def text_table(header, rows):
    widths = [max(len(str(x)) for x in col) + 2 for col in zip(header, *rows)]
    line = '+' + '+'.join('-' * w for w in widths) + '+'
    fmt = lambda r: '|' + '|'.join(str(x).center(w) for x, w in zip(r, widths)) + '|'
    return '\n'.join([line, fmt(header), line] + [fmt(r) for r in rows] + [line])


def summarize(results_root, metric):
    table = np.load(os.path.join(results_root, 'results.npz'))[metric]
    per_video = table.mean(axis=1)
    return per_video.mean(), per_video.std() / np.sqrt(per_video.size)


def main(args=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('dest_path', type=str)
    parser.add_argument('--results', type=str, nargs='+', required=True, help='quant_results_root:label pairs')
    parser.add_argument('--mean_precision', type=int, default=2)
    parser.add_argument('--std_err_precision', type=int, default=3)
    args = parser.parse_args(args)
    os.makedirs(args.dest_path, exist_ok=True)
    pairs = [r.rsplit(':', 1) if ':' in r else (r, os.path.basename(r.rstrip('/'))) for r in args.results]
    for metric, mp, sp in (('psnr', args.mean_precision, args.std_err_precision),
                           ('ssim', args.mean_precision + 2, args.std_err_precision + 3)):
        rows = []
        for root, label in pairs:
            mean, err = summarize(root, metric)
            rows.append([label, '%.*f' % (mp, mean), '%.*f' % (sp, err)])
        with open(os.path.join(args.dest_path, '%s_perf_summary.txt' % metric), 'w') as f:
            f.write(text_table(['Model', 'Mean', 'StdErr'], rows))
        print(text_table(['Model', 'Mean', 'StdErr'], rows))


if __name__ == '__main__':
    main()
