#!/bin/bash
# round-5 baseline on one box: bench line, per-layer table, ATen sources
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_base
mkdir -p $OUT
cd $ROOT
python3 bench.py --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
EAS_LT_SORT=time EAS_LT_ROWS=400 python3 scripts/layer_times.py 2 > $OUT/layer_times.txt 2>&1
python3 scripts/dev_aten_sources.py 2 > $OUT/aten_sources.txt 2>&1
tail -5 $OUT/aten_sources.txt
