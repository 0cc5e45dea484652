#!/bin/bash
# development (GPU box): SQ / TCC counters + dispatch info of the weight-gradient kernel on one row of scripts/dev_wgrad_shapes.py
# usage: pmc_wgrad_shape.sh <s|m> <row> <tag>
W=${1:-m}; ROW=${2:-15}; TAG=${3:-pmc_wg}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAVE_READY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
            "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $PASS --kernel-include-regex "conv_wgrad_mfma" --output-format csv -d $OUT/$N -- python3 $ROOT/scripts/dev_wgrad_shapes.py $W 1 $ROW > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$OUT/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        info = {}
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'][40:110]
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
            info[k] = {c: row.get(c) for c in ('Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'Scratch_Size')}
        for k, cs in acc.items():
            print(k, info[k])
            for c, v in cs.items():
                print(f'   {c:28s} n={len(v)} mean={sum(v)/len(v):.5g}')
PY
