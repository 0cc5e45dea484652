#!/bin/bash
# A/B of an environment setting on configs 3, 4, 5 (batch 32): usage r05_cfg_ab.sh "<env A>" "<env B>"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_cfg_ab
mkdir -p $OUT
cd $ROOT
for c in ${3:-3 4 5}; do
  for v in A B; do
    if [ $v = A ]; then E="$1"; else E="$2"; fi
    env $E EAS_BENCH_GRAPH=1 EAS_BENCH_NO_EVAL=1 timeout 600 python3 bench.py --config $c --no-cpu-baseline > $OUT/bench_c${c}_$v.log 2>&1
    echo "config $c [$E] rc=$? $(tail -1 $OUT/bench_c${c}_$v.log | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])
except Exception as e: print('parse error', e)")"
  done
done
