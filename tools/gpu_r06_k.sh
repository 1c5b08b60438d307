#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "area_attn or a2c2f" 2>&1 | tail -2
for i in 1 2 3; do QKV_SHAPE=20 python tools/micro/attn_qkv_ab.py 2>&1 | grep -v amdgpu.ids; done
python tools/micro/attn_qkv_ab.py 2>&1 | grep -v amdgpu.ids
run() { YMK_DISABLE=$2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['config']['value_sync'], r['ms_per_step'], r['op_calls_per_step'], {k: v['ms_per_step_eager'] for k, v in r['roofline_layers'].items()})"; }
for i in 1 2 3; do run fused 0; run unfused 2097152; done
