#!/bin/bash
# same-box A/B/C... of environment settings on config 2: usage r05_env_ab.sh "<env1>;<env2>;..." [reps]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_env_ab
mkdir -p $OUT
cd $ROOT
IFS=';' read -ra SETS <<< "$1"
for r in $(seq 1 ${2:-2}); do
  i=0
  for E in "${SETS[@]}"; do
    i=$((i+1))
    env $E EAS_BENCH_GRAPH=1 EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1 EAS_BENCH_NO_EMA=1 timeout 300 python3 bench.py --no-cpu-baseline > $OUT/bench_${i}_$r.log 2>&1
    echo "[$E] rc=$? $(tail -1 $OUT/bench_${i}_$r.log | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])
except Exception as e: print('parse error', e)")"
  done
done
