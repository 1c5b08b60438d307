#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic of the bench workload.
# Usage: tools/gpu_profile.sh <tag>      (outputs under gpurun_out/<tag>_*)
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
set -e
python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --split 1 --pipeline 1 --no-sync-leg > /dev/null   # refuse to profile a crashing workload
set +e
BENCH="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-graph --split 1 --pipeline 1 --no-sync-leg"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace -- $BENCH > $R/gpurun_out/${TAG}_trace.log 2>&1
python $R/tools/prof_summary.py $(ls $R/gpurun_out/${TAG}_trace/*/*.db | head -1) 15 > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$C -- $BENCH > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
  python $R/tools/pmc_summary.py $(ls $R/gpurun_out/${TAG}_pmc_$C/*/*.db | head -1) > $R/gpurun_out/${TAG}_pmc_$C.json 2>&1
  rm -rf $R/gpurun_out/${TAG}_pmc_$C
done
rm -rf $R/gpurun_out/${TAG}_trace
head -40 $R/gpurun_out/${TAG}_kernel_stats.txt
head -c 600 $R/gpurun_out/${TAG}_pmc_FETCH_SIZE.json
