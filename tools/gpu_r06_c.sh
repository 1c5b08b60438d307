#!/bin/bash
# round 6, call C: the 256 x 208 tile of the LDS-DMA core — tests, per-call times with / without, bench A/B (YMK_GLDS_TILE208)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_next.py -q -x -k "glds" 2>&1 | tail -4
for v in 1 0; do
  YMK_GLDS_TILE208=$v YMK_BENCH_CALLS=gpurun_out/r06c_calls_t208_$v.log python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-sync-leg > /dev/null 2>&1
  echo "--- YMK_GLDS_TILE208=$v: $(head -1 gpurun_out/r06c_calls_t208_$v.log)"
  grep -E "conv_glds_kernel<256" gpurun_out/r06c_calls_t208_$v.log | sort -k1,1n | cut -c1-150
done
bash tools/micro/env_ab.sh YMK_GLDS_TILE208 "1 0" 2 "conv_glds"
