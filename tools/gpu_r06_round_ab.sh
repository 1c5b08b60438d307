#!/bin/bash
# Same-box interleaved A/B of the round: the round-5 tree (git worktree of the round-5 verdict commit, built, staged under .r05tree for this call only) against this tree
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { ( cd $2 && python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null ) | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', 'value', r['value'], 'sync', r['value_sync'], 'ms', r['ms_per_step'], 'calls', r['op_calls_per_step'])"; }
for i in 1 2 3 4; do run r05 $R/.r05tree; run r06 $R; done
