#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r03i}
timeout -k 10 900 python -m pytest tests/test_gpu_next.py tests/test_gpu_kernels.py tests/test_gpu_mixture.py -m gpu -q --tb=short --no-header -p no:cacheprovider -k "glds or conv or cat2 or expert or attention or attn or a2c2f or config5" > gpurun_out/${T}_tests.log 2>&1
echo "tests: exit $?"; tail -3 gpurun_out/${T}_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/${T}_tests.log | head -20
timeout -k 10 600 python tools/micro/glds_tile_ab3.py 64 > gpurun_out/${T}_glds_tile_ab_b64.txt 2>&1; echo "ab: exit $?"; cat gpurun_out/${T}_glds_tile_ab_b64.txt
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/${T}_bench.json 2>/dev/null
python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench.json').read()); print('bench:', r['value'], r['ms_per_step'])
for f in r['families'][:12]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_cfg5.json 2> gpurun_out/${T}_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench_cfg5.json').read()); print('cfg5:', r['value'], r['ms_per_step'])
for f in r['families'][:10]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
