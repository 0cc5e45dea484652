#!/bin/bash
# development (GPU box): rocprofv3 kernel stats of the eager step for two settings of one environment switch, side by side.
# usage: ab_kernel_stats.sh VAR A B [config]
set -u
VAR=$1; A=$2; B=$3; CFG=${4:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_$VAR
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (side figures off: the 640x640 one runs in a child process the profiler would follow, writing a second kernel_stats.csv)
export EAS_BENCH_GRAPH=0 EAS_BENCH_NO_640=1 EAS_BENCH_NO_EMA=1
for v in $A $B; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_$v -- python3 $ROOT/bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline > $OUT/log_$v.txt 2>&1
  find $OUT/s_$v -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$v.csv \;
  rm -rf $OUT/s_$v
done
python3 - <<PY
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return d
a, b = load('$OUT/kernel_stats_$A.csv'), load('$OUT/kernel_stats_$B.csv')
import re
def fam(n):
    return re.sub(r'\(.*', '', n)[:90]
rows = []
for n in set(a) | set(b):
    ca, ta = a.get(n, (0, 0.0)); cb, tb = b.get(n, (0, 0.0))
    rows.append((ta - tb, n, ca, ta, cb, tb))
rows.sort(key=lambda r: -abs(r[0]))
print('total ms: $VAR=$A', round(sum(v[1] for v in a.values()), 2), ' $VAR=$B', round(sum(v[1] for v in b.values()), 2))
for d, n, ca, ta, cb, tb in rows[:45]:
    print(f'{d:+8.2f} ms  A {ca:5d} calls {ta:8.2f} ms | B {cb:5d} calls {tb:8.2f} ms  {fam(n)}')
PY
