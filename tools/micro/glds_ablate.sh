#!/bin/bash
# Stage ablation of the LDS-DMA tiled convolution core (csrc/conv_glds.hip, -DGLDS_ABLATE=<bits>): one library per variant, built in the
# build container (`tools/micro/glds_ablate.sh build`, under tools/micro/_dwab/, git-ignored), timed on the GPU box (`tools/micro/glds_ablate.sh`).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "${1:-}" = build ]; then
  for v in 0 1 2 4 8 3 7 16 19 23; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Iyolo_master_amd/csrc -DGLDS_ABLATE=$v yolo_master_amd/csrc/conv_glds.hip -o tools/micro/_dwab/libglds_$v.so 2>/dev/null &
  done
  wait; ls tools/micro/_dwab | grep glds; exit 0
fi
python - <<'PY'
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from yolo_master_amd import _lib, ops
bf = torch.bfloat16
p = lambda t: C.c_void_p(t.data_ptr())
names = {0: "baseline", 8: "no MFMA", 1: "no fragment reads, no MFMA", 2: "no DMA after the prologue", 4: "no barrier", 3: "no DMA, no compute", 7: "loop skeleton",
         16: "no output stores", 19: "no DMA, compute or stores", 23: "skeleton without stores", 100: "baseline again"}
shapes = [(64, 64, 3, 1, 40), (128, 64, 3, 1, 80), (256, 256, 3, 2, 80), (128, 128, 3, 2, 160), (384, 256, 1, 1, 40), (256, 768, 1, 1, 20)]
st = torch.cuda.current_stream().cuda_stream
print("us per call (two-stage / three-stage loop), batch 64, bf16")
print(f"{'variant':30s}" + "".join(f"{'%d->%d k%d s%d @%d' % sh:>22s}" for sh in shapes))
data = {}
for cin, cout, k, s, hw in shapes:
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(64, hw, hw, cin, generator=g).to(bf).cuda()
    w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, generator=g) * (k * k * cin) ** -0.5, bf).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    ho = (hw + 2 * (k // 2) - k) // s + 1
    y = torch.empty((64, ho, ho, cout), dtype=bf, device="cuda")
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, 64, hw, hw, cin, cout, k, s, cin, cout, 0, w.shape[1], _lib.ACT_SILU)
    data[(cin, cout, k, s, hw)] = (x, w, bias, y, d)
for v, nm in names.items():
    lib = C.CDLL(f"tools/micro/_dwab/libglds_{0 if v == 100 else v}.so")
    row = []
    for sh in shapes:
        x, w, bias, y, d = data[sh]
        cell = []
        for two in (1, 0):
            call = lambda: lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), two, C.c_void_p(st))
            for _ in range(40): call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): rc = call()
            e1.record(); torch.cuda.synchronize()
            assert rc == 0, rc
            cell.append(e0.elapsed_time(e1) * 100)
        row.append(f"{cell[0]:9.1f} /{cell[1]:7.1f}")
    print(f"{nm:30s}" + "".join(f"{c:>22s}" for c in row))
PY
