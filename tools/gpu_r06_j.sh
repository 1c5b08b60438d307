#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3; do
  python tools/micro/attn_ab.py yolo_master_amd/libymk.so "tree" 2>&1 | grep -v amdgpu.ids
  python tools/micro/attn_ab.py tools/micro/_dwab/libymk_atprio.so "setprio" 2>&1 | grep -v amdgpu.ids
  python tools/micro/attn_qkv_ab.py 2>&1 | grep -v amdgpu.ids; python tools/micro/attn_qkv_ab.py tools/micro/_dwab/libymk_atprio.so "setprio" 2>&1 | grep -v amdgpu.ids
done
