#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r05_tests}
mkdir -p $OUT
cd $ROOT
if [ -n "${2:-}" ]; then
  timeout 3000 python3 -m pytest tests -m gpu -q -k "$2" > $OUT/gpu_tests.log 2>&1
else
  timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
fi
echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/gpu_tests.log | tail -40
