#!/bin/bash
# development (GPU box): rocprofv3 kernel stats of the eval-mode forward (scripts/dev_eval.py), per forward
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-2}
OUT=$ROOT/gpurun_out/prof_eval
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EAS_DEV_EVAL_ONLY=auto
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 $ROOT/scripts/dev_eval.py $CFG > $OUT/log.txt 2>&1
find $OUT/s -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_config$CFG.csv \;
rm -rf $OUT/s
tail -5 $OUT/log.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/kernel_stats_config$CFG.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:45]:
    print('%8.3f ms %6s calls avg %7.1f us  %s' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:150]))
PY
