#!/bin/bash
# same-box A/B: bench step with two libraries / env settings.  usage: r05_ab.sh <tag> "<env A>" "<env B>" [reps]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p $OUT
cd $ROOT
REPS=${4:-2}
for r in $(seq 1 $REPS); do
  for v in A B; do
    if [ $v = A ]; then E="$2"; else E="$3"; fi
    env $E EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1 timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_${v}_$r.log 2>&1
    tail -1 $OUT/bench_${v}_$r.log > $OUT/bench_${v}_$r.json.log
    python3 - <<PY
import json
d=json.loads(open('$OUT/bench_${v}_$r.json.log').read())
f=d['roofline']['hip_kernel_ms_per_step']
print('$v $r [$E]', d['ms_per_step'], 'conv_fwd', f['eas_conv_fwd']['ms_per_step'], 'wgrad', f['eas_conv_wgrad']['ms_per_step'])
PY
  done
done
