#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -k "nms" 2>&1 | tail -3
bash tools/micro/calls_ab.sh " nms " 2 before=tools/micro/_dwab/libymk_nms_before.so
