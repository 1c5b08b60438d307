#!/bin/bash
# BUILD HOST: stage-ablated copies of the depthwise stencil (csrc/dwconv.hip -DDW_ABLATE bits: 1 no global loads, 2 no LDS staging writes,
# 4 one filter row instead of K, 8 no output stores) as standalone libraries; GPU box: python tools/micro/dw_ablate_ab.py
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_dwab
for a in 0 1 3 4 8 15; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDW_ABLATE=$a yolo_master_amd/csrc/dwconv.hip -o tools/micro/_dwab/libdw_abl$a.so 2>/dev/null &
done
wait
ls tools/micro/_dwab/libdw_abl*.so
