#!/bin/bash
# GPU box: ablation of the matrix-core depthwise kernel's item loop (csrc/dwmfma.hip, -DDWM_ABLATE=<bits>): one shared library per
# variant, the same shape timed through each.   Usage: tools/micro/dw_ablate.sh [H C k]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
H=${1:-160}; C=${2:-128}; K=${3:-9}
mkdir -p /tmp/dwab
for v in 0 1 2 4 8 16 3 12 31; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDWM_ABLATE=$v yolo_master_amd/csrc/dwmfma.hip -o /tmp/dwab/libdw_$v.so 2>/dev/null &
done
wait
python - "$H" "$C" "$K" <<'PY'
import ctypes as C, sys, torch
H, Cc, k = map(int, sys.argv[1:4])
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
x = torch.randn(64, H, H, Cc, generator=g).to(bf).cuda()
w = (torch.randn(k * k, Cc, generator=g) / k).to(bf).cuda()
out = torch.empty_like(x)
vp = lambda t: C.c_void_p(t.data_ptr())
names = {0: "baseline", 1: "no global loads", 2: "no LDS staging writes", 4: "no MFMA", 8: "no global stores", 16: "no output-tile writes",
         3: "no loads + no staging", 12: "no MFMA + no stores", 31: "empty loop"}
for v, nm in names.items():
    lib = C.CDLL(f"/tmp/dwab/libdw_{v}.so")
    lib.ymk_dw_toeplitz_elems.restype = C.c_size_t
    tp = torch.empty(lib.ymk_dw_toeplitz_elems(Cc, k), dtype=bf, device="cuda")
    lib.ymk_dw_toeplitz_pack(vp(w), Cc, k, vp(tp), None)
    s = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.ymk_dwconv2d_mfma(vp(x), vp(tp), None, None, vp(out), 64, H, H, Cc, k, Cc, Cc, 0, 0, C.c_void_p(s))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"  {nm:28s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
PY
