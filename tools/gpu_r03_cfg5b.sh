#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_mixture.py tests/test_gpu_kernels.py -m gpu -q --tb=short --no-header -p no:cacheprovider -k "attention or attn or a2c2f or norms or router or config5" > gpurun_out/r03e_tests.log 2>&1
echo "tests: exit $?"; tail -3 gpurun_out/r03e_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/r03e_tests.log | head -20
for V in 0 524288; do
YMK_DISABLE=$V python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03e_bench_cfg5_$V.json 2> gpurun_out/r03e_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/r03e_bench_cfg5_$V.json').read()); print('cfg5 YMK_DISABLE=$V:', r['value'], r['ms_per_step'])
for f in r['families'][:14]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
done
