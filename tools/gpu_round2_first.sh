#!/bin/bash
# First GPU call of the next round, in one box: (1) the validated suite still green, (2) first hardware run of the
# config-5 kernels, (3) the GEMM design probes, (4) a bench line.  Usage (from the build container):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round2_first.sh r02'
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_suite.log 2>&1; echo "gpu suite exit $?"; tail -3 gpurun_out/${TAG}_gpu_suite.log
bash tools/gpu_cfg5.sh $TAG | tail -60
bash tools/gpu_probe.sh $TAG | tail -60
for e in 1 3; do   # A/B of the opt-in core inside the same box: 1 = three LDS stages, 3 = two
  YMK_ENABLE=$e timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_enable$e.json 2>/dev/null
  echo "YMK_ENABLE=$e: $(python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_enable$e.json'));print(d['value'], d['ms_per_step'])" 2>/dev/null)"
done
# BASELINE config 5 (L-scale moa-mot model, 1280 px, 16 images) on the first-implementation kernels: a first number, not a bench line
YMK_EXPERIMENTAL=1 timeout -k 10 400 python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline \
  > gpurun_out/${TAG}_bench_cfg5.json 2> gpurun_out/${TAG}_bench_cfg5.err; echo "config-5 bench exit $?"; head -c 600 gpurun_out/${TAG}_bench_cfg5.json; echo
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"; cat gpurun_out/${TAG}_bench.json
