#!/bin/bash
# Runs on the GPU box (via gpurun): the round's profile set.
#   1. FETCH_SIZE calibration (scripts/micro/fetch_calib, 4/8/16-byte loads over 1 GiB)
#   2. configs 2 and 3: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (scripts/gpu_profile.sh)
#   3. configs 4, 5: rocprofv3 kernel stats of the eager step
#   4. config 2: SQ counters of the dense-convolution kernels (scripts/pmc_conv_step.sh)
# Results: gpurun_out/<tag>/...
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- $ROOT/scripts/micro/fetch_calib > $OUT/calib.log 2>&1
python3 $ROOT/scripts/pmc_summary.py --calibrate $OUT/calib $OUT/fetch_calibration.json > $OUT/fetch_calibration.txt 2>&1
cat $OUT/fetch_calibration.txt
bash $ROOT/scripts/gpu_profile.sh $TAG/config2 $OUT/fetch_calibration.json 2
bash $ROOT/scripts/gpu_profile.sh $TAG/config3 $OUT/fetch_calibration.json 3
export EAS_BENCH_GRAPH=0 EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c$c -- python3 $ROOT/bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_c$c.log 2>&1
  find $OUT/stats_c$c -name '*kernel_stats.csv' -exec cp {} $OUT/config${c}_kernel_stats.csv \;
  rm -rf $OUT/stats_c$c
  tail -1 $OUT/bench_c$c.log | cut -c1-200
done
bash $ROOT/scripts/pmc_conv_step.sh $TAG/conv_sq > $OUT/conv_sq.log 2>&1
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
