#!/bin/bash
# Run on the GPU box (via gpurun): SQ counter passes over the matrix-core depthwise kernel on one shape.
# Usage: tools/gpu_dw_pmc.sh <tag> <H> <C> <k>      (outputs gpurun_out/<tag>_dwpmc_<n>.json)
set -u
TAG=$1; H=$2; C=$3; K=$4
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
n=0
for CTR in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$((n+1))
  timeout -k 10 150 rocprofv3 --pmc $CTR --kernel-trace -d $R/gpurun_out/${TAG}_dwpmc_$n -- python $R/tools/gpu_dw_one.py $H $C $K mfma > $R/gpurun_out/${TAG}_dwpmc_$n.log 2>&1
  python $R/tools/pmc_summary.py $(ls $R/gpurun_out/${TAG}_dwpmc_$n/*/*.db | head -1) dw_mfma > $R/gpurun_out/${TAG}_dwpmc_$n.json 2>&1
  rm -rf $R/gpurun_out/${TAG}_dwpmc_$n
  cat $R/gpurun_out/${TAG}_dwpmc_$n.json | python -c "
import json,sys
d=json.load(sys.stdin)['kernels']
for k,v in d.items():
    print(k[:40], {c: round(x['mean']) for c,x in v.items()})"
done
