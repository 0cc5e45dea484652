#!/bin/bash
# usage (on the GPU box): pmc_kernel.sh <kernel regex> <tag> ["script args.."] -- SQ counters of one kernel family under one eager
# bench.py step (or another driver script with its arguments, path relative to the repo root)
export EAS_BENCH_GRAPH=0
RX=${1:-smallconv}; TAG=${2:-pmc_k}; SCRIPT=${3:-bench.py --steps 1 --warmup 1 --no-cpu-baseline}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_INSTS_WAVE32_LDS SQ_WAVE_READY"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $PASS --kernel-include-regex "$RX" --output-format csv -d $OUT/$N -- python3 $ROOT/$SCRIPT > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$OUT/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row['Kernel_Name'][:70]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in acc.items():
            print(k)
            for c, v in cs.items():
                print(f'   {c:28s} n={len(v)} mean={sum(v)/len(v):.4g}')
PY
