#!/bin/bash
# round 5: grouped head launches -- unit + whole-head parity, then the bench step with and without them on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_group
mkdir -p $OUT
cd $ROOT
timeout 600 python3 scripts/dev_group.py 64 > $OUT/dev_group.txt 2>&1
echo "dev_group rc=$?"
tail -25 $OUT/dev_group.txt
for g in 0 1; do
  EAS_HEAD_GROUP=$g EAS_BENCH_NO_EVAL=1 timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_g$g.log 2>&1
  echo "group=$g rc=$? $(tail -1 $OUT/bench_g$g.log | cut -c1-330)"
done
