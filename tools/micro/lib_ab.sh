#!/bin/bash
# Interleaved A/B of two builds of libymk.so on one GPU box: tools/micro/lib_ab.sh <other.so> [grep pattern for the per-call log]
# (the tree's own library = "new"; the other one is swapped in under the package for its runs and the tree's library restored at the end)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OTHER=$1; PAT=${2:-}
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['value_sync'], r['ms_per_step'])"; }
cp yolo_master_amd/libymk.so /tmp/libymk_new.so
for i in 1 2 3; do
  cp /tmp/libymk_new.so yolo_master_amd/libymk.so; run new
  cp $OTHER yolo_master_amd/libymk.so; run other
done
if [ -n "$PAT" ]; then
  YMK_BENCH_CALLS=/tmp/calls_other.log python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-sync-leg > /dev/null 2>&1
  cp /tmp/libymk_new.so yolo_master_amd/libymk.so
  YMK_BENCH_CALLS=/tmp/calls_new.log python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-sync-leg > /dev/null 2>&1
  echo "--- other"; head -1 /tmp/calls_other.log; grep "$PAT" /tmp/calls_other.log | sort -k1,1n | cut -c1-150
  echo "--- new"; head -1 /tmp/calls_new.log; grep "$PAT" /tmp/calls_new.log | sort -k1,1n | cut -c1-150
fi
cp /tmp/libymk_new.so yolo_master_amd/libymk.so
