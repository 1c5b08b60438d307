#!/bin/bash
# Kernel trace of the TIMED configuration (captured graph, two 32-image sub-batches on two streams): 200 replayed steps next to 5 + 2
# eager sub-batched steps and 3 whole-batch eager steps of the diagnostic leg — the averages are the in-graph kernels' to ~5 %.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_gtrace -- python $R/bench.py --steps 200 --warmup 5 --split 2 --pipeline 1 --no-cpu-baseline > $R/gpurun_out/r03_gtrace.log 2>&1
echo "trace: exit $?"; tail -2 $R/gpurun_out/r03_gtrace.log | cut -c1-300
( echo "# rocprofv3 --kernel-trace of bench.py --steps 200 --warmup 5 --split 2 --pipeline 1 (hipGraph replays: kernels of 32-image sub-batches on two streams);"
  echo "# per-step columns divide by 210 steps (200 replays + 5 warm-up + 2 + 3 whole-batch eager steps of the diagnostic leg)"
  python $R/tools/prof_summary.py $(ls $R/gpurun_out/r03_gtrace/*/*.db | head -1) 210 ) > $R/gpurun_out/r03_graph_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/r03_gtrace
head -30 $R/gpurun_out/r03_graph_kernel_stats.txt
