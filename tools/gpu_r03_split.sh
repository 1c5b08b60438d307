#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for S in 1 2 4 1 2; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --split $S > gpurun_out/r03f_bench_s$S.json 2> gpurun_out/r03f_bench_s$S.err
  python -c "
import json
r=json.loads(open('gpurun_out/r03f_bench_s$S.json').read()); print('split $S:', r['value'], r['ms_per_step'], r['config']['launch'])" || tail -5 gpurun_out/r03f_bench_s$S.err
done
python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03f_bench_cfg5.json 2> gpurun_out/r03f_bench_cfg5.err
python -c "
import json
r=json.loads(open('gpurun_out/r03f_bench_cfg5.json').read()); print('cfg5:', r['value'], r['ms_per_step'])
for f in r['families'][:8]: print('   ', f['kernel'], f['ms_per_step'], f['launches_per_step'], f['frac'], f['achieved_tflops'])"
