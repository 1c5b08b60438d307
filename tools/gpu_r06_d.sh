#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for r in 1 2; do for v in 0 1 2 3; do YMK_MLP_NW8=$v python tools/micro/mlp_ab.py "YMK_MLP_NW8=$v" 2>&1 | grep -v amdgpu.ids; done; done
