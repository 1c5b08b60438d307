#!/bin/bash
# GPU box: per-call times (YMK_BENCH_CALLS: median of three eager steps per op call) and the bench line's value / value_sync for the tree's
# libymk.so ("tree") and for each variant library given (tools/micro/lib_variant.sh), one bench.py process per library, ROUNDS rounds interleaved.
#   tools/micro/calls_ab.sh "<grep -E pattern of op calls>" <rounds> name1=path1.so [name2=path2.so ...]      (extra bench flags: $YMK_AB_FLAGS)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
PAT=$1; ROUNDS=$2; shift 2
cp yolo_master_amd/libymk.so /tmp/libymk_tree.so
LIBS="tree=/tmp/libymk_tree.so $@"
for r in $(seq 1 $ROUNDS); do
  for kv in $LIBS; do
    n=${kv%%=*}; f=${kv#*=}
    cp $f yolo_master_amd/libymk.so
    YMK_BENCH_CALLS=/tmp/calls_${n}_$r.log python bench.py --steps 40 --warmup 10 --no-cpu-baseline $YMK_AB_FLAGS 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('round $r', '$n', 'value', r['value'], 'sync', r['value_sync'], 'ms', r['ms_per_step'])"
  done
done
cp /tmp/libymk_tree.so yolo_master_amd/libymk.so
for kv in $LIBS; do
  n=${kv%%=*}
  echo "--- $n (round 1 per-call log: $(head -1 /tmp/calls_${n}_1.log))"
  grep -E "$PAT" /tmp/calls_${n}_1.log | sort -k1,1n | cut -c1-150
done
