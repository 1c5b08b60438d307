#!/bin/bash
# round 6, call A: attention stage ablation / two-query-tile A/B, and the bench line with the new record fields
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
bash tools/micro/attn_ab.sh > gpurun_out/r06_attn_ab.txt 2>&1
echo skip
echo skip
cat gpurun_out/r06_attn_ab.txt
