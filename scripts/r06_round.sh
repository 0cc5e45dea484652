#!/bin/bash
# round 6 (GPU box, via gpurun): the bench lines of configs 2-5 (config 2 with every side figure), the per-layer parity figures
# (tests/parity_report.py), and N runs of the one-rank RCCL bench in the record-then-rendezvous order.
# usage: r06_round.sh <tag> [rccl runs: 10]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
RUNS=${2:-10}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python3 bench.py > $OUT/bench.log 2> $OUT/bench.err
echo "bench rc=$?"
tail -1 $OUT/bench.log > $OUT/bench.json.log
python3 - <<PY
import json
d=json.loads(open('$OUT/bench.json.log').read())
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'launches', d.get('launches_per_step'))
print('weight_average', d.get('weight_average'))
print('parity', d.get('parity'))
print('roofline frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline'].get('traffic_source', '')[:80])
print('eval', d.get('eval_forward_frames_per_s'))
print('canvas_640', {k: v for k, v in (d.get('canvas_640') or {}).items() if k != 'hip_kernel_ms_per_step'})
print('cpu', d.get('cpu_baseline'))
PY
python3 tests/parity_report.py $OUT/parity_teacher_forced.json > $OUT/parity_report.log 2>&1
echo "parity_report rc=$? $(tail -1 $OUT/parity_report.log | cut -c1-400)"
for c in 3 4 5; do
  EAS_BENCH_NO_EVAL=0 timeout 900 python3 bench.py --config $c > $OUT/bench_config$c.log 2> $OUT/bench_config$c.err
  echo "config $c rc=$? $(tail -1 $OUT/bench_config$c.log | cut -c1-260)"
  tail -1 $OUT/bench_config$c.log > $OUT/bench_config$c.json.log
done
ok=0
for i in $(seq 1 $RUNS); do
  EAS_BENCH_FORCE_DDP=1 EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1 EAS_BENCH_NO_EMA=1 MASTER_PORT=$((29600+i)) timeout 300 python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline > $OUT/rccl_$i.log 2>&1
  rc=$?
  [ $rc -eq 0 ] && ok=$((ok+1))
  echo "rccl run $i rc=$rc $(tail -1 $OUT/rccl_$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'], d['config']['rccl_ranks'], d['config']['launch'][:40])" 2>/dev/null)"
done
echo "one-rank RCCL runs clean: $ok of $RUNS"
