#!/bin/bash
# Run on the GPU box (via gpurun): first hardware run of the config-5 kernels (include/ymk_mixture.h) and of the opt-in
# entry points of include/ymk_next.h.  One pytest process per test group, so that a faulting kernel only takes its own
# group down; no -x: every failure of a group is listed.   Usage: tools/gpu_cfg5.sh [tag]
set -u
TAG=${1:-cfg5}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
LOG=gpurun_out/${TAG}_cfg5.log
: > $LOG
run() {   # run <label> <pytest args...>
  echo "=== $1" >> $LOG
  YMK_EXPERIMENTAL=1 timeout -k 10 300 python -m pytest "${@:2}" -m gpu -q --tb=short --no-header -p no:cacheprovider >> $LOG 2>&1
  echo "$1: exit $? — $(tail -1 $LOG)"
}
run norms      tests/test_gpu_mixture.py -k test_norms_and_elementwise
run pools      tests/test_gpu_mixture.py -k test_pools_stats_shuffle_gather
run routers    tests/test_gpu_mixture.py -k test_router_tails
run attention  tests/test_gpu_mixture.py -k test_attention_family
run modules    tests/test_gpu_mixture.py -k test_modules_vs_reference_golden
run model      tests/test_gpu_mixture.py -k test_config5_model_vs_reference_golden
run glds       tests/test_gpu_next.py -k "glds_direct"
run post       tests/test_gpu_next.py -k "scale_boxes or segment or process_mask"
run glds_model tests/test_gpu_next.py -k test_model_with_the_new_core_enabled
grep -E "^(FAILED|ERROR)|Error|error:" $LOG | head -40
