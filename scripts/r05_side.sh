#!/bin/bash
# sweep of the side-stream batching of the weight-gradient slab kernels: "count:us" pairs
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_side
mkdir -p $OUT
cd $ROOT
for v in ${1:-"0:0 16:0 0:0 16:0"}; do
  n=${v%%:*}; us=${v##*:}
  AT=""; case "$n" in *,*) AT=$n; n=0;; esac
  EAS_WGRAD_SIDE_AT=$AT EAS_WGRAD_SIDE=$n EAS_WGRAD_SIDE_US=$us EAS_BENCH_GRAPH=1 EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1 timeout 300 python3 bench.py --no-cpu-baseline > $OUT/bench_${n}_${AT}_${us}.log 2>&1
  echo "side=$n at=$AT us=$us rc=$? $(tail -1 $OUT/bench_${n}_${AT}_${us}.log | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])
except Exception as e: print('parse error', e)")"
done
