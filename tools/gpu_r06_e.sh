#!/bin/bash
# round 6, call E (record): the dot2 depthwise stencil with the next tile's halo prefetched into registers (DW_PAIRDOT_PF; the source with that path is
# parked: tools/micro/parked/dwconv_pairdot_pf.hip.txt) at several run lengths
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
cp yolo_master_amd/libymk.so /tmp/libymk_tree.so
one() {  # name lib target
  cp $2 yolo_master_amd/libymk.so
  YMK_DW_WG_TARGET=$3 YMK_BENCH_CALLS=/tmp/calls_$1.log python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', 'value', r['value'], 'sync', r['value_sync'], {k: v['ms_per_step_eager'] for k, v in r['roofline_layers'].items()})"
  grep -E "moe_dw|dwconv " /tmp/calls_$1.log | sort -k1,1n | awk '{printf "   %s %s %s %s %s %s | %s us\n", $2,$3,$4,$5,$6,$7,$(NF-5)}' | head -8
}
for r in 1 2; do
  one tree /tmp/libymk_tree.so 0
  one pf_default tools/micro/_dwab/libymk_dwpf.so 0
  one pf_t4096 tools/micro/_dwab/libymk_dwpf.so 4096
  one pf_t2048 tools/micro/_dwab/libymk_dwpf.so 2048
  one pf_t16384 tools/micro/_dwab/libymk_dwpf.so 16384
done
cp /tmp/libymk_tree.so yolo_master_amd/libymk.so
