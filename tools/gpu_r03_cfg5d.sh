#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r03o}
timeout -k 10 900 python -m pytest tests/test_gpu_mixture.py -m gpu -q --tb=short --no-header -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "tests: exit $?"; tail -3 gpurun_out/${T}_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/${T}_tests.log | head -20
for i in 1 2; do
python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_cfg5_$i.json 2> gpurun_out/${T}_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench_cfg5_$i.json').read()); print('cfg5:', r['value'], r['ms_per_step'])
for f in r['families'][:14]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
done
