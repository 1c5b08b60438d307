#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r03h}
timeout -k 10 300 tools/micro/fillrate.bin > gpurun_out/${T}_fillrate.txt 2>&1; echo "fillrate: exit $?"; cat gpurun_out/${T}_fillrate.txt
timeout -k 10 600 python tools/micro/glds_tile_ab3.py 64 > gpurun_out/${T}_glds_tile_ab_b64.txt 2>&1; echo "ab: exit $?"; cat gpurun_out/${T}_glds_tile_ab_b64.txt
