#!/bin/bash
# round 6, call B: stage ablation of the projection phase of the fused qkv + attention kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python tools/micro/attn_qkv_ab.py 2>&1 | grep -v amdgpu.ids
for a in 1 2 4 16 17 18 20 23; do python tools/micro/attn_qkv_ab.py tools/micro/_dwab/libymk_qa$a.so "ablate $a" 2>&1 | grep -v amdgpu.ids; done
python tools/micro/attn_qkv_ab.py 2>&1 | grep -v amdgpu.ids
