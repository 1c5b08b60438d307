#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python tools/micro/glds_tile_ab3.py 64 > gpurun_out/r03c_glds_tile_ab.txt 2>&1; echo "ab: exit $?"; cat gpurun_out/r03c_glds_tile_ab.txt
for W in 1 2 3; do
  YMK_GLDS_BIG_MIN_WAVES=$W python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r03c_bench_w$W.json 2>/dev/null
  python -c "
import json
r=json.loads(open('gpurun_out/r03c_bench_w$W.json').read()); print('min waves $W:', r['value'], r['ms_per_step'])
for f in r['families'][:8]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
done
YMK_GLDS_BIG_MIN_WAVES=1000 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r03c_bench_nobig.json 2>/dev/null
python -c "
import json
r=json.loads(open('gpurun_out/r03c_bench_nobig.json').read()); print('no big tiles:', r['value'], r['ms_per_step'])"
python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03c_bench_cfg5.json 2> gpurun_out/r03c_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/r03c_bench_cfg5.json').read()); print('cfg5:', r['value'], r['ms_per_step'])
for f in r['families'][:24]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
