#!/bin/bash
# Round-5 second GPU call: moe_pw prefetch distance A/B, Detect class kernel stage ablation (per-call times), kernel tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "esmoe or detect" > gpurun_out/r05b_kernel_tests.log 2>&1; echo "kernel tests: exit $?"; tail -2 gpurun_out/r05b_kernel_tests.log
bash tools/micro/calls_ab.sh "moe_pw" 2 pwahead1=tools/micro/_dwab/libymk_pwahead1.so > gpurun_out/r05b_pw_ab.txt 2>&1
grep -E "^round|^---|moe_pw" gpurun_out/r05b_pw_ab.txt
export YMK_AB_FLAGS="--no-graph --steps 2 --warmup 2 --no-sync-leg --pipeline 1"
L=""; for v in 1 2 4 8 16 32 63; do L="$L dcab$v=tools/micro/_dwab/libymk_dcab$v.so"; done
bash tools/micro/calls_ab.sh "detect_cls" 1 $L > gpurun_out/r05b_dc_ablate.txt 2>&1
grep -E "^---|detect_cls" gpurun_out/r05b_dc_ablate.txt
