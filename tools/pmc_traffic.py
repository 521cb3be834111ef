#!/usr/bin/env python
"""profiles/traffic.json from the PMC passes of a round.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> \\
        --tag r02b --command "python bench.py --steps 3 --warmup 1 --no-graph ..." [--out profiles/traffic.json]

The two directories are rocprofv3 outputs of the SAME command, one counter each (FETCH_SIZE costs 3
of the 4 TCC slots, WRITE_SIZE 2: they do not fit one pass; collected with --kernel-trace only, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  Per hand-written kernel (namespace mtr::, keyed
by the function name without template arguments, launches of all instantiations averaged):

    bytes = FETCH_SIZE [KiB] x 1024 x 2 + WRITE_SIZE [KiB] x 1024

The x 2 is the guide's gfx950 correction: this rocprofv3 tallies the 128-byte requests of wide
coalesced streaming reads at 64 B.  It is calibrated on the decode kernel (1.2836 GB of logits read
exactly once: FETCH_SIZE x 2 = algorithmic bytes to 0.15 %) and the pyramid kernel (uint8 frames
read once).  Gather-type reads (the sampler's 8-byte taps) are uncalibrated: the file carries the
raw counters so that either reading can be reconstructed.
bench.py reads this file for `roofline.traffic`.
"""
import argparse
import csv
import glob
import json
import os
import re
import time
from collections import defaultdict


def base_name(kernel):
    m = re.search(r'mtr::(\w+)', kernel)
    return m.group(1) if m else None


def per_kernel_average(directory, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row['Counter_Name'] != counter:
                    continue
                name = base_name(row['Kernel_Name'])
                if name:
                    acc[name].append(float(row['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('--tag', required=True)
    ap.add_argument('--command', default='')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  'profiles', 'traffic.json'))
    a = ap.parse_args()
    fetch = per_kernel_average(a.fetch_dir, 'FETCH_SIZE')
    write = per_kernel_average(a.write_dir, 'WRITE_SIZE')
    out = {
        '_source': f'{a.tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (two passes) of '
                   f'`{a.command}`, reduced by tools/pmc_traffic.py on {time.strftime("%Y-%m-%d")}',
        '_doc': 'HBM-side bytes per launch (average over the launches of the pass, all template '
                'instantiations of a kernel together): bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950: wide '
                'streaming reads are tallied at half their size, MI355X_MICROARCH.md) + WRITE_SIZE KiB x 1024; '
                'raw counters alongside.  Reads served by the Infinity Cache are counted, not excluded.',
    }
    for name in sorted(set(fetch) | set(write)):
        f_kib, n_f = fetch.get(name, (0.0, 0))
        w_kib, n_w = write.get(name, (0.0, 0))
        out[name] = dict(bytes=int(round(f_kib * 1024 * 2 + w_kib * 1024)), fetch_KiB_raw=round(f_kib, 1),
                         write_KiB=round(w_kib, 1), fetch_correction=2, launches_in_pass=max(n_f, n_w))
    with open(a.out, 'w') as f:
        json.dump(out, f, indent=1)
    print(f'wrote {a.out}: {len(out) - 2} kernels')


if __name__ == '__main__':
    main()
