#!/bin/bash
# Round-5 GPU call c: new kernel paths (proj+MLP, attention wave counts, chunked ES-MoE, fused-vs-unfused decode), A/B by environment.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -k "mlp or esmoe or attn or a2c2f or ablock or fused_decode or detect or nms or dw" > gpurun_out/r05c_tests.log 2>&1; echo "tests: exit $?"; tail -4 gpurun_out/r05c_tests.log
bash tools/micro/env_ab.sh YMK_ATTN_WAVES "4 5 6" 2 "area_attn" > gpurun_out/r05c_attn_waves.txt 2>&1; cat gpurun_out/r05c_attn_waves.txt
bash tools/micro/env_ab.sh YMK_DISABLE "0 8388608" 2 "mlp_fused|conv1x1_ws|conv_glds_kernel<128, 2, 128>" > gpurun_out/r05c_projmlp.txt 2>&1; cat gpurun_out/r05c_projmlp.txt
