#!/bin/bash
# Run on the GPU box (via gpurun): the design probes for the next tiled-GEMM core, each self-checking against a naive
# kernel and timed next to the library's current kernels.  Builds what is missing (hipcc is on the box), bounded by
# timeouts.  Usage: tools/gpu_probe.sh [tag]      (logs under gpurun_out/<tag>_*.log)
set -u
TAG=${1:-probe}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for p in conv256 gemm256; do
  if [ ! -x tools/micro/$p.bin ] || [ tools/micro/$p.hip -nt tools/micro/$p.bin ]; then
    timeout 300 hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/micro/$p.hip -o tools/micro/$p.bin \
      > gpurun_out/${TAG}_build_$p.log 2>&1 || { echo "build of $p failed"; tail -5 gpurun_out/${TAG}_build_$p.log; }
  fi
done
timeout -k 5 120 tools/micro/conv256.bin > gpurun_out/${TAG}_conv256.log 2>&1; echo "conv256 exit $?"
timeout -k 5 120 tools/micro/gemm256.bin > gpurun_out/${TAG}_gemm256.log 2>&1; echo "gemm256 exit $?"
cat gpurun_out/${TAG}_conv256.log
tail -8 gpurun_out/${TAG}_gemm256.log
timeout -k 5 240 python tools/gpu_diag.py glds > gpurun_out/${TAG}_glds_ab.log 2>&1; echo "glds per-shape A/B exit $?"; cat gpurun_out/${TAG}_glds_ab.log
