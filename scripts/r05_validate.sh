#!/bin/bash
# round 5: GPU tests, the bench line with its new side figures, five runs of the one-rank RCCL bench (the capture-abort chase)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05_val}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
echo "pytest rc=$?"
tail -15 $OUT/gpu_tests.log
timeout 900 python3 bench.py --no-cpu-baseline > $OUT/bench.log 2>&1
echo "bench rc=$?"
tail -1 $OUT/bench.log > $OUT/bench.json.log
python3 - <<PY
import json
d=json.loads(open('$OUT/bench.json.log').read())
print('ms_per_step', d['ms_per_step'], 'launches', d.get('launches_per_step'))
print('canvas_640', d.get('canvas_640'))
r=d.get('roofline_eval') or {}
print('roofline_eval', {k:v for k,v in r.items() if k!='hip_kernel_ms_per_batch'})
print('eval', d.get('eval_forward_frames_per_s'))
PY
for i in 1 2 3 4 5; do
  EAS_BENCH_FORCE_DDP=1 EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1 MASTER_PORT=$((29600+i)) timeout 300 python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline > $OUT/rccl_$i.log 2>&1
  echo "rccl run $i rc=$? $(tail -1 $OUT/rccl_$i.log | cut -c1-120)"
done
