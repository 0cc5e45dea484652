#!/bin/bash
# usage (on the GPU box): pmc_one.sh <shape idx> <tag> ; SQ counters of the conv kernel for one bench shape
IDX=${1:-1}; TAG=${2:-pmc_one}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -f $ROOT/gpurun_out/counters.txt ] || rocprofv3 -L > $ROOT/gpurun_out/counters.txt 2>&1
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $PASS --kernel-include-regex "conv_fwd_mfma" --output-format csv -d $OUT/$N -- python3 $ROOT/scripts/dev_conv.py one $IDX > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$OUT/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            print(f'{k:36s} n={len(v)} last={v[-1]:.4g} mean={sum(v)/len(v):.4g}')
    for f in glob.glob(d + '**/*kernel_trace.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        if rows:
            r = rows[-1]
            print('  kernel', r['Kernel_Name'][:60], 'dur_us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'grid', r.get('Grid_Size_X', r.get('Grid_Size')), 'vgpr', r.get('VGPR_Count'), 'lds', r.get('LDS_Block_Size'))
PY
