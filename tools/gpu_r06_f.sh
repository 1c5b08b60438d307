#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-sync-leg "$@" 2>/tmp/err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$*', '->', r['value'], r['ms_per_step'])" || tail -3 /tmp/err.txt; }
for r in 1 2; do
  run --pipeline 3
  run --pipeline 2
  run --pipeline 2 --cu-mask
  run --pipeline 3 --cu-mask
  run --pipeline 4 --cu-mask
done
