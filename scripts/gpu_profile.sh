#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of bench.py plus two PMC passes (FETCH_SIZE / WRITE_SIZE in
# separate passes, kernel-trace only) for the hand-written HBM-bound kernels.  Results land in gpurun_out/<tag>/.
set -u
TAG=${1:-prof}
CFG=${3:-2}
export EAS_BENCH_NO_640=1    # (the 640x640 side figure runs in a child process that rocprofv3 would follow)
export EAS_BENCH_NO_EVAL=1   # the training step alone (the eval side figure replays its own graphs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EAS_BENCH_GRAPH=0   # profile eager launches: one row per kernel dispatch (graph replay is what bench.py times by default)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name '*kernel_trace.csv' -delete
RX='bn_lif|bn_stats|bn_silu|lif_fwd|lif_bwd|arsnn|event_hist|smallconv|conv_|conv1x1|conv3x3'
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python3 $ROOT/scripts/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json ${2:-} > $OUT/pmc_traffic.txt 2>&1
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
tail -3 $OUT/bench_under_rocprof.log; head -30 $OUT/pmc_traffic.txt
