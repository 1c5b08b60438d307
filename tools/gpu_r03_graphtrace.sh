#!/bin/bash
# Kernel trace of the TIMED configuration (bench.py defaults: captured graphs, three whole batches in flight on three streams): 200 replayed
# steps next to the warm-up / capture steps and 3 whole-batch eager steps of the diagnostic leg — the averages are the in-graph kernels' to ~5 %.
# Usage: tools/gpu_r03_graphtrace.sh [extra bench.py flags, e.g. "--split 2 --pipeline 1"] -> gpurun_out/r03_graph_kernel_stats.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
FLAGS=${1:-}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_gtrace -- python $R/bench.py --steps 200 --warmup 6 $FLAGS --no-cpu-baseline > $R/gpurun_out/r03_gtrace.log 2>&1
echo "trace: exit $?"; grep -o '"value": [0-9.]*, "unit": "images/sec"' $R/gpurun_out/r03_gtrace.log | head -1
( echo "# rocprofv3 --kernel-trace of bench.py --steps 200 --warmup 6 $FLAGS (hipGraph replays);"
  echo "# per-step columns divide by 212 steps (200 replays + 6 warm-up + 3 first replays + 3 whole-batch eager steps of the diagnostic leg)"
  python $R/tools/prof_summary.py $(ls $R/gpurun_out/r03_gtrace/*/*.db | head -1) 212 ) > $R/gpurun_out/r03_graph_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/r03_gtrace
head -34 $R/gpurun_out/r03_graph_kernel_stats.txt
