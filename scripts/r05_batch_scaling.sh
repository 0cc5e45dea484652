#!/bin/bash
# per-family kernel time at batch 16 / 32 / 64: what does not scale with the batch is per-launch fixed cost
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r05_scaling}
mkdir -p $OUT
cd $ROOT
for b in 16 32 64; do
  EAS_BENCH_NO_EVAL=1 EAS_BENCH_NO_640=1 timeout 600 python3 bench.py --no-cpu-baseline --batch $b > $OUT/bench_b$b.log 2>&1
  tail -1 $OUT/bench_b$b.log > $OUT/bench_b$b.json.log
done
python3 - <<PY
import json
d={b:json.loads(open('$OUT/bench_b%d.json.log'%b).read()) for b in (16,32,64)}
print('step ms', {b:d[b]['ms_per_step'] for b in d}, 'launches', d[64]['launches_per_step'])
f={b:d[b]['roofline']['hip_kernel_ms_per_step'] for b in d}
print('%-28s %6s %8s %8s %8s %8s'%('family','calls','b16','b32','b64','fixed'))
tot=0
for k in sorted(f[64], key=lambda k:-f[64][k]['ms_per_step']):
    a,b,c=(f[x].get(k,{}).get('ms_per_step',0) for x in (16,32,64))
    fixed=(4*a-c)/3
    tot+=fixed
    print('%-28s %6d %8.3f %8.3f %8.3f %8.3f'%(k,f[64][k]['calls']//3,a,b,c,fixed))
print('sum of fixed parts', round(tot,3))
PY
