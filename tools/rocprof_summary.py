#!/usr/bin/env python
"""Reduce rocprofv3 CSV output to a small, commit-able summary.

    python tools/rocprof_summary.py <rocprof_output_dir> <summary.md> [--ours-only]

* kernel trace (``*_kernel_trace.csv``): per-kernel count / total / average / min / max duration;
* counter collection (``*_counter_collection.csv``): per-kernel average of each PMC counter.
Hand-written kernels of libmetrabs_hip.so live in namespace ``mtr::``.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=110):
    name = name.replace('void ', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = [f'# rocprofv3 summary of `{os.path.basename(src.rstrip("/"))}`', '']
    traces = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    for path in traces:
        agg = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                agg[row['Kernel_Name']].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
        total = sum(sum(v) for v in agg.values())
        lines += [f'## kernel trace ({os.path.basename(path)}): {sum(len(v) for v in agg.values())} '
                  f'dispatches, {total / 1e6:.3f} ms of kernel time', '',
                  '| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
        ranked = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
        shown = 0
        for name, d in ranked:
            ours = 'mtr::' in name
            if shown >= 25 and not ours:
                continue
            shown += 1
            lines.append(f'| {"**" if ours else ""}{short(name)}{"**" if ours else ""} | {len(d)} | '
                         f'{sum(d) / 1e6:.3f} | {sum(d) / len(d) / 1e3:.2f} | {min(d) / 1e3:.2f} | '
                         f'{max(d) / 1e3:.2f} | {100 * sum(d) / total:.2f} |')
        ours_total = sum(sum(d) for n, d in agg.items() if 'mtr::' in n)
        lines += ['', f'hand-written (mtr::) kernels: {ours_total / 1e6:.3f} ms = '
                      f'{100 * ours_total / max(total, 1):.2f} % of kernel time', '']
    for path in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        with open(path) as f:
            for row in csv.DictReader(f):
                agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        lines += [f'## PMC counters ({os.path.basename(path)}), per-dispatch averages', '',
                  '| kernel | counter | dispatches | average | min | max |', '|---|---|---|---|---|---|']
        for name, counters in sorted(agg.items()):
            if 'mtr::' not in name:
                continue
            for cname, vals in sorted(counters.items()):
                lines.append(f'| {short(name)} | {cname} | {len(vals)} | {sum(vals) / len(vals):.1f} | '
                             f'{min(vals):.1f} | {max(vals):.1f} |')
        lines.append('')
    with open(dst, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print(f'wrote {dst} ({len(lines)} lines)')


if __name__ == '__main__':
    main()
