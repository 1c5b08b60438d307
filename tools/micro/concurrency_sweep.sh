#!/bin/bash
# bench.py --pipeline P --split S / --sync-split S sweep on one box (interleaved; images/s, value_sync)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$*', '->', r['value'], r['value_sync'], r['ms_per_step'], r['p50_batch_ms_sync'])"; }
for rep in 1 2; do
run --pipeline 3 --split 1 --sync-split 2
run --pipeline 5 --split 1 --sync-split 2
run --pipeline 4 --split 1 --sync-split 4
run --pipeline 2 --split 2 --sync-split 1
run --pipeline 3 --split 2 --sync-split 2
run --pipeline 6 --split 1 --sync-split 2
done
