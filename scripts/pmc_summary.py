#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md prescribes) into
per-kernel HBM traffic per launch.  gfx950 corrections from the guide: both counters are in KiB; FETCH_SIZE reads
exactly half of the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact.

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'[<(].*', '', name)


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                a = acc[short(row['Kernel_Name'])]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    return acc


def main():
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        fetch_b = 2.0 * 1024.0 * f / max(nf, 1)
        write_b = 1024.0 * w / max(nw, 1)
        out[k] = {'launches': max(nf, nw), 'fetch_bytes_per_launch': round(fetch_b), 'write_bytes_per_launch': round(write_b),
                  'hbm_bytes_per_launch': round(fetch_b + write_b),
                  'note': 'FETCH_SIZE KiB x2 (gfx950 wide-read correction) + WRITE_SIZE KiB'}
    with open(sys.argv[3], 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for k, v in out.items():
        print(f"{k:40s} n={v['launches']:5d} fetch {v['fetch_bytes_per_launch'] / 1e6:9.3f} MB write {v['write_bytes_per_launch'] / 1e6:9.3f} MB")


if __name__ == '__main__':
    main()
