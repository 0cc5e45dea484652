#!/usr/bin/env python3
"""Every kernel family of a profiled step against the HBM time of its MEASURED traffic: rocprofv3 kernel stats (average duration) beside the
PMC bytes per launch (scripts/gpu_profile.sh -> pmc_traffic.json) at the 6.3 TB/s a streaming copy reaches on MI355X.
usage: kernel_vs_traffic.py [kernel_stats.csv] [pmc_traffic.json] [steps in the stats run: 11]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    stats = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r04_v6_kernel_stats.csv')
    pmc = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'profiles', 'r04_v6_pmc_traffic.json')
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 11
    pm = json.load(open(pmc))
    fam = {}
    for r in csv.DictReader(open(stats)):
        n = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        k = re.match(r'(\w+)', n).group(1)
        d = fam.setdefault(k, [0.0, 0])
        d[0] += float(r['TotalDurationNs'])
        d[1] += int(r['Calls'])
    out = []
    for k, (ns, calls) in fam.items():
        if k in pm and calls:
            avg_us = ns / calls / 1e3
            hb = pm[k]['hbm_bytes_per_launch']
            ideal_us = hb / 6.3e12 * 1e6
            out.append((ns / steps / 1e6 * (1 - ideal_us / avg_us), k, calls / steps, avg_us, hb / 1e6, ideal_us))
    print('  above    kernel family                        calls   average   traffic     at 6.3 TB/s')
    for ex, k, c, avg, mb, ideal in sorted(out, reverse=True):
        print(f'{ex:7.3f} ms  {k:36s} {c:5.1f} {avg:8.1f} us {mb:9.1f} MB {ideal:8.1f} us  x{avg / ideal if ideal > 0.05 else float("nan"):.1f}')


if __name__ == '__main__':
    main()
