#!/bin/bash
# BUILD HOST: the depthwise stencil (csrc/dwconv.hip) in its four forms (DW_LDS_MODE 0 / 1 / 2 / 3, see the file's header) as standalone
# libraries under tools/micro/_dwab/ (git-ignored, travels with the gpurun snapshot).  GPU box: python tools/micro/dw_lds_ab.py
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_dwab
for m in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDW_LDS_MODE=$m yolo_master_amd/csrc/dwconv.hip -o tools/micro/_dwab/libdw_mode$m.so 2>/dev/null &
done
wait
ls -la tools/micro/_dwab/libdw_mode*.so
