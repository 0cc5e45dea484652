#!/bin/bash
# usage (on the GPU box): pmc_conv_step.sh <tag> [bench args] -- SQ counters of every dense-convolution kernel of ONE eager bench.py step
# (two passes of 8 SQ counters + GRBM_GUI_ACTIVE, kernel-trace only), summarised per kernel symbol by scripts/pmc_conv_summary.py
export EAS_BENCH_GRAPH=0 EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1
TAG=${1:-pmc_conv}; shift
ARGS=${*:---steps 1 --warmup 1 --no-cpu-baseline}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RX='conv_fwd_mfma|conv1x1_mfma|conv_wgrad_mfma|conv1x1_wgrad|conv_dgrad_s2|conv3x3_group|conv1x1_group|conv_wgrad_group'
I=0
for PASS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  I=$((I+1))
  rocprofv3 --kernel-trace --pmc $PASS --kernel-include-regex "$RX" --output-format csv -d $OUT/pass$I -- python3 $ROOT/bench.py $ARGS > $OUT/pass$I.log 2>&1
done
python3 $ROOT/scripts/pmc_conv_summary.py $OUT $OUT/conv_sq_counters.json > $OUT/conv_sq_counters.txt 2>&1
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
head -40 $OUT/conv_sq_counters.txt
