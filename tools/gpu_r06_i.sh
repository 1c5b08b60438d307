#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "area_attn or a2c2f" 2>&1 | tail -2
for i in 1 2 3; do python tools/micro/attn_qkv_ab.py 2>&1 | grep -v amdgpu.ids; python tools/micro/attn_qkv_ab.py tools/micro/_dwab/libymk_qkvw0.so "W per pass" 2>&1 | grep -v amdgpu.ids; done
