#!/bin/bash
# round-5 profile set: the round script (calibration, configs 2-5 stats, PMC traffic configs 2/3, SQ counters) + the eval forward's kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash scripts/gpu_profile_round.sh ${1:-r05} > gpurun_out/profile_round.log 2>&1
tail -5 gpurun_out/profile_round.log
bash scripts/prof_eval.sh 2 > gpurun_out/prof_eval.log 2>&1
tail -30 gpurun_out/prof_eval.log
