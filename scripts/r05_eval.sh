#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_eval
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_model.py -m gpu -q -k "eval_head_with_grouped or eval_forward or fuse_model or evaluator" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
for g in 0 1; do
  EAS_HEAD_GROUP=$g EAS_BENCH_NO_640=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > $OUT/bench_g$g.log 2>&1
  echo "group=$g rc=$? $(tail -1 $OUT/bench_g$g.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['eval_forward_frames_per_s']['value'], d['eval_forward_frames_per_s']['fuse_model'], d['roofline_eval']['hip_kernels_ms_per_batch'])")"
done
