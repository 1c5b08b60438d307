#!/bin/bash
# Round 3: half k-steps of the LDS-DMA convolution core (A/B per shape + end to end), the restructured matrix-core attention and the
# XCD-aware GroupNorm statistics (config 5), parity tests of what changed.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r03f}
timeout -k 10 900 python -m pytest tests/test_gpu_mixture.py tests/test_gpu_kernels.py tests/test_gpu_next.py -m gpu -q --tb=short --no-header -p no:cacheprovider -k "attention or attn or a2c2f or norm or glds or config5 or conv" > gpurun_out/${T}_tests.log 2>&1
echo "tests: exit $?"; tail -3 gpurun_out/${T}_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/${T}_tests.log | head -20
timeout -k 10 600 python tools/micro/glds_tile_ab3.py 64 > gpurun_out/${T}_glds_tile_ab.txt 2>&1; echo "ab: exit $?"; cat gpurun_out/${T}_glds_tile_ab.txt
for V in 0 1 3 5; do
  YMK_GLDS_BK32=$V python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/${T}_bench_k32_$V.json 2>/dev/null
  python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench_k32_$V.json').read()); print('YMK_GLDS_BK32=$V:', r['value'], r['ms_per_step'])
for f in r['families'][:8]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
done
for V in 0 1; do
YMK_GLDS_BK32=$V python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_cfg5_$V.json 2> gpurun_out/${T}_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/${T}_bench_cfg5_$V.json').read()); print('cfg5 YMK_GLDS_BK32=$V:', r['value'], r['ms_per_step'])
for f in r['families'][:16]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
done
