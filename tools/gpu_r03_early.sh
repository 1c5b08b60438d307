#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py tests/test_gpu_bench_flow.py -m gpu -q --tb=short --no-header -p no:cacheprovider > gpurun_out/r03i_tests.log 2>&1; echo "tests: exit $?"; tail -3 gpurun_out/r03i_tests.log | cut -c1-200; grep -E "^(FAILED|ERROR)" gpurun_out/r03i_tests.log | head
for V in "0 0" "1048576 0" "0 16" "0 0" "1048576 0"; do
  set -- $V
  YMK_DISABLE=$1 YMK_ENABLE=$2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r03i_b.json 2> gpurun_out/r03i_b.err
  python -c "
import json
r=json.loads(open('gpurun_out/r03i_b.json').read()); print('DISABLE=$1 ENABLE=$2:', r['value'], r['ms_per_step'])" || tail -3 gpurun_out/r03i_b.err
done
YMK_BENCH_SPLIT=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r03i_b.json 2>/dev/null; python -c "
import json
r=json.loads(open('gpurun_out/r03i_b.json').read()); print('split 1, early levels:', r['value'], r['ms_per_step'])"
YMK_BENCH_SPLIT=1 YMK_DISABLE=1048576 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r03i_b.json 2>/dev/null; python -c "
import json
r=json.loads(open('gpurun_out/r03i_b.json').read()); print('split 1, no early levels:', r['value'], r['ms_per_step'])"
