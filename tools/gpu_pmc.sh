#!/bin/bash
# Run on the GPU box (via gpurun): SQ counter passes over one eager forward+NMS of the bench workload.
# Usage: tools/gpu_pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]   (outputs gpurun_out/<tag>_sq_<n>.json)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --split 1 --pipeline 1 --no-sync-leg"
n=0
for C in "$@"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/${TAG}_sq_$n -- $BENCH > $R/gpurun_out/${TAG}_sq_$n.log 2>&1
  python $R/tools/pmc_summary.py $(ls $R/gpurun_out/${TAG}_sq_$n/*/*.db | head -1) > $R/gpurun_out/${TAG}_sq_$n.json 2>&1
  rm -rf $R/gpurun_out/${TAG}_sq_$n
done
