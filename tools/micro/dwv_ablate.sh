#!/bin/bash
# Stage ablation of the VALU depthwise kernel (csrc/dwconv.hip, -DDW_ABLATE=<bits>): one shared library per variant, built in the
# build container (`tools/micro/dwv_ablate.sh build`, libraries under tools/micro/_dwab/, git-ignored) and timed on the GPU box
# (`tools/micro/dwv_ablate.sh [H C]`): 64 x H x H x C bf16, k = 3, 5, 7, 9, every variant.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "${1:-}" = build ]; then
  for v in 0 1 2 4 8 3 12 15; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Iyolo_master_amd/csrc -DDW_ABLATE=$v yolo_master_amd/csrc/dwconv.hip -o tools/micro/_dwab/libdw_$v.so 2>/dev/null &
  done
  wait; ls tools/micro/_dwab; exit 0
fi
H=${1:-160}; C=${2:-128}
python - "$H" "$C" <<'PY'
import ctypes as C, sys, torch
H, Cc = map(int, sys.argv[1:3])
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
x = torch.randn(64, H, H, Cc, generator=g).to(bf).cuda()
out = torch.empty_like(x)
vp = lambda t: C.c_void_p(t.data_ptr())
names = {0: "baseline", 1: "no global loads", 2: "no LDS staging writes", 4: "one filter row only", 8: "no global stores",
         3: "no loads + no staging", 12: "one row + no stores", 15: "skeleton"}
print(f"64 x {H} x {H} x {Cc} bf16; us per call")
print(f"{'variant':28s}" + "".join(f"{'k=%d' % k:>10s}" for k in (3, 5, 7, 9)))
for v, nm in names.items():
    lib = C.CDLL(f"tools/micro/_dwab/libdw_{v}.so")
    row = []
    for k in (3, 5, 7, 9):
        w = (torch.randn(k * k, Cc, generator=g) / k).to(bf).cuda()
        s = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.ymk_dwconv2d(1, vp(x), vp(w), None, None, vp(out), 64, H, H, Cc, k, Cc, Cc, 0, 0, C.c_void_p(s))
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): rc = call()
        e1.record(); torch.cuda.synchronize()
        assert rc == 0, rc
        row.append(e0.elapsed_time(e1) * 100)
    print(f"{nm:28s}" + "".join(f"{t:10.1f}" for t in row))
PY
