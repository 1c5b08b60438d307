#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r03n}
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_next.py -m gpu -q --tb=short --no-header -p no:cacheprovider -k "attn or attention or mlp or conv or model" > gpurun_out/${T}_tests.log 2>&1
echo "tests: exit $?"; tail -3 gpurun_out/${T}_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/${T}_tests.log | head -20
for i in 1 2; do
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/${T}_bench_$i.json 2>/dev/null
python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench_$i.json').read()); print('bench:', r['value'], r['ms_per_step'])
for f in r['families'][:26]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
done
