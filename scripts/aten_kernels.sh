#!/bin/bash
# development (GPU box): the kernels of the eager training step that are NOT ours (ATen glue, copies, fills), per step, from a rocprofv3 kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/aten
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (side figures off: the 640x640 one runs in a child process the profiler would follow, writing a second kernel_stats.csv)
export EAS_BENCH_GRAPH=0 EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1 EAS_BENCH_NO_EMA=1
STEPS=${1:-6}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 $ROOT/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline > $OUT/log.txt 2>&1
find $OUT/s -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/s
python3 - <<PY
import csv, re
rows = list(csv.DictReader(open('$OUT/kernel_stats.csv')))
steps = $STEPS + 3 + 3           # timed + warm-up + the three event-timed steps
ours = ('anonymous namespace', 'eas_')
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps
other = [r for r in rows if not any(o in r['Name'] for o in ours)]
print(f'all kernels {tot:.2f} ms/step; not ours {sum(float(r["TotalDurationNs"]) for r in other) / 1e6 / steps:.3f} ms/step in {sum(int(r["Calls"]) for r in other) / steps:.0f} launches/step')
for r in sorted(other, key=lambda r: -float(r['TotalDurationNs']))[:40]:
    nm = re.sub(r'^void ', '', r['Name'])[:170]
    print('%7.3f ms/step %6.1f calls/step avg %6.1f us  %s' % (float(r['TotalDurationNs']) / 1e6 / steps, int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, nm))
PY
