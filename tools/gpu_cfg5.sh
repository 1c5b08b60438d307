#!/bin/bash
# Run on the GPU box (via gpurun): first hardware run of the config-5 kernels (include/ymk_mixture.h).
# Usage: tools/gpu_cfg5.sh [tag]    (log under gpurun_out/<tag>_cfg5.log)
set -u
TAG=${1:-cfg5}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
YMK_EXPERIMENTAL=1 timeout -k 10 600 python -m pytest tests/test_gpu_mixture.py -m gpu -q -x --no-header -p no:cacheprovider \
  > gpurun_out/${TAG}_cfg5.log 2>&1
echo "exit $?"
tail -40 gpurun_out/${TAG}_cfg5.log
