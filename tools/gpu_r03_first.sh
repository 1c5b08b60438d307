#!/bin/bash
# Round 3, first hardware call: config 5 at its own configuration (L scale, 1280^2), the bench line of the round-2 tree on this box,
# SQ counter passes for the kernels of that tree.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mixture.py -m gpu -q -k "own_configuration" -s --no-header -p no:cacheprovider > gpurun_out/r03a_cfg5_l.log 2>&1
echo "cfg5_l: exit $?"; grep -E "config 5|passed|failed|Error|assert" gpurun_out/r03a_cfg5_l.log | head -20
python bench.py --steps 30 --warmup 10 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo "bench: exit $?"; head -c 400 gpurun_out/r03a_bench.json; echo
bash tools/gpu_pmc.sh r03a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE"
ls -la gpurun_out/ | grep r03a
head -c 1500 gpurun_out/r03a_sq_1.json
