#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r03k}
timeout -k 10 900 python -m pytest tests/test_gpu_mixture.py tests/test_gpu_next.py -m gpu -q --tb=short --no-header -p no:cacheprovider -k "norm or config5 or modules or glds or conv" > gpurun_out/${T}_tests.log 2>&1
echo "tests: exit $?"; tail -3 gpurun_out/${T}_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/${T}_tests.log | head -20
YMK_BENCH_CALLS=gpurun_out/${T}_cfg5_calls.log python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_cfg5.json 2> gpurun_out/${T}_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench_cfg5.json').read()); print('cfg5:', r['value'], r['ms_per_step'])
for f in r['families'][:40]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
head -70 gpurun_out/${T}_cfg5_calls.log
