#!/bin/bash
# Round 3: the whole GPU suite on the current tree + the bench line in the three compute types.   Usage: tools/gpu_r03_suite.sh <tag>
set -u
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -q --tb=short --no-header -p no:cacheprovider -s > gpurun_out/${TAG}_gpu_suite.log 2>&1
echo "suite: exit $?"; tail -3 gpurun_out/${TAG}_gpu_suite.log; grep -E "^(FAILED|ERROR)" gpurun_out/${TAG}_gpu_suite.log | head -40
grep -E "^config [235]|flipped" gpurun_out/${TAG}_gpu_suite.log | head -60
for DT in bf16 f16 f32; do
  python bench.py --steps 30 --warmup 10 --dtype $DT $( [ $DT != bf16 ] && echo --no-cpu-baseline ) > gpurun_out/${TAG}_bench_$DT.json 2> gpurun_out/${TAG}_bench_$DT.err
  echo "bench $DT: exit $?"; python -c "
import json,sys
r=json.loads(open('gpurun_out/${TAG}_bench_$DT.json').read()); print(r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['frac'], r.get('cpu_baseline'))" 2>&1 | tail -1
done
