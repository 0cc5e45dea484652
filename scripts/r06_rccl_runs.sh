#!/bin/bash
# round 6 (GPU box): N consecutive runs of the one-rank RCCL bench (record the graphs, THEN create the process group); every run must exit 0
# with rccl_ranks == 1, three graph replays per step and the same final loss.  usage: r06_rccl_runs.sh <tag> [N: 100]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06_rccl}
N=${2:-100}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
ok=0
: > $OUT/summary.txt
for i in $(seq 1 $N); do
  EAS_BENCH_FORCE_DDP=1 EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1 EAS_BENCH_NO_EMA=1 MASTER_PORT=$((29600 + i % 300)) timeout 300 python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline > $OUT/run.log 2> $OUT/run.err
  rc=$?
  line=$(tail -1 $OUT/run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'], d['config']['rccl_ranks'], d['config']['launch'][:32])" 2>/dev/null)
  echo "run $i rc=$rc $line" >> $OUT/summary.txt
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else cp $OUT/run.err $OUT/failed_$i.err; fi
done
echo "one-rank RCCL runs clean: $ok of $N; distinct final losses: $(awk '{print $5}' $OUT/summary.txt | sort -u | wc -l)" | tee -a $OUT/summary.txt
tail -3 $OUT/summary.txt
