#!/usr/bin/env python3
"""bench.py's cpu_baseline leg (the oracle on a bounded sample of the configuration-2 workload) at several torch thread counts on this host.
usage: cpu_baseline_sweep.py [threads ...]   (no GPU needed; one fresh process per thread count)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]
code = ("import sys, json; sys.path.insert(0, %r); import bench; from eas_snn_amd import workloads; "
        "print('RESULT ' + json.dumps(bench.cpu_baseline(workloads.get(2), 8, 200000)))" % ROOT)
for n in counts:
    env = dict(os.environ, EAS_CPU_THREADS=str(n), OMP_NUM_THREADS=str(n))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('RESULT ')]
    if not line:
        print(f'threads {n}: failed', out.stderr[-300:])
        continue
    r = json.loads(line[0][7:])
    print(f"threads {n:4d}: {r['value']:8.3f} event-frames/s   ({r['cpu_model']}, {r['host_logical_cpus']} logical CPUs; {r['sample']})", flush=True)
