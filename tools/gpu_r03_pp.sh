#!/bin/bash
# Round 3: buffer-resource staging + out-of-phase loop of the LDS-DMA convolution core: parity tests, per-shape A/B (batch 64 and 32), end to end.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r03g}
timeout -k 10 900 python -m pytest tests/test_gpu_next.py tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short --no-header -p no:cacheprovider -k "glds or conv or model or cat2 or expert" > gpurun_out/${T}_tests.log 2>&1
echo "tests: exit $?"; tail -3 gpurun_out/${T}_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/${T}_tests.log | head -20
for B in 64 32; do
timeout -k 10 600 python tools/micro/glds_tile_ab3.py $B > gpurun_out/${T}_glds_tile_ab_b$B.txt 2>&1; echo "ab: exit $?"; cat gpurun_out/${T}_glds_tile_ab_b$B.txt
done
for V in 0 1 16 48 17; do
  YMK_GLDS_PP=$V python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/${T}_bench_pp_$V.json 2>/dev/null
  python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench_pp_$V.json').read()); print('YMK_GLDS_PP=$V:', r['value'], r['ms_per_step'])
for f in r['families'][:8]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
done
for V in 0 16; do
YMK_GLDS_PP=$V python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_cfg5_$V.json 2> gpurun_out/${T}_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench_cfg5_$V.json').read()); print('cfg5 YMK_GLDS_PP=$V:', r['value'], r['ms_per_step'])
for f in r['families'][:6]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
done
