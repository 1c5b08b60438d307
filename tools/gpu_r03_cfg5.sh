#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_mixture.py tests/test_gpu_next.py -m gpu -q --tb=short --no-header -p no:cacheprovider > gpurun_out/r03d_mixture.log 2>&1
echo "mixture suite: exit $?"; tail -3 gpurun_out/r03d_mixture.log; grep -E "^(FAILED|ERROR)" gpurun_out/r03d_mixture.log | head -20
python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03d_bench_cfg5.json 2> gpurun_out/r03d_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/r03d_bench_cfg5.json').read()); print('cfg5:', r['value'], r['ms_per_step'])
for f in r['families'][:26]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r03d_bench.json 2>/dev/null
python -c "
import json
r=json.loads(open('gpurun_out/r03d_bench.json').read()); print('S bf16:', r['value'], r['ms_per_step'])
for f in r['families'][:10]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
