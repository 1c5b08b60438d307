#!/bin/bash
# GPU box: stage ablation of the fused ES-MoE kernel (csrc/esfused.hip, -DESF_ABLATE=<bits>): one shared library per variant, timed on the
# S detector's layer shapes at batch 64 with the bench's routing density.  bits: 1 no stencil arithmetic, 2 no halo transfers, 4 no MFMAs,
# 8 no epilogue, 16 no filter-slice / weight loads.   usage: tools/micro/esf_ablate.sh [variants...]
cd "$(dirname "$0")/../.."
mkdir -p /tmp/esfab
VARS="${@:-0 1 2 4 8 16 6 14 30 31}"
for v in $VARS; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DESF_ABLATE=$v yolo_master_amd/csrc/esfused.hip -o /tmp/esfab/libesf_$v.so 2>/dev/null &
done
wait
python - $VARS <<'PY'
import ctypes as C, sys, torch
torch.manual_seed(0)
dev = "cuda"
bf = torch.bfloat16
def vp(t): return C.c_void_p(t.data_ptr())
shapes = [(128, 160, 83), (256, 80, 77), (256, 40, 117)]
for Cc, HW, pairs in shapes:
    B, E, top_k, ks = 64, 4, 2, [3, 5, 7, 9]
    x = torch.randn(B, HW, HW, Cc, device=dev).to(bf)
    parts, offs, off = [], [], 0
    for k in ks:
        w = (torch.randn(k * k, Cc, device=dev) / k).to(bf); parts.append(w.reshape(-1)); offs.append(off); off += w.numel()
    dw_w = torch.cat(parts); dw_off = torch.tensor(offs, dtype=torch.int32, device=dev); ksz = torch.tensor(ks, dtype=torch.int32, device=dev)
    pw_w = (torch.randn(E, Cc, Cc, device=dev) * Cc ** -0.5).to(bf)
    pw_b = torch.randn(E, Cc, device=dev) * 0.3; ns = 1 + 0.1 * torch.randn(Cc, device=dev); nt = 0.2 * torch.randn(Cc, device=dev)
    sel = torch.full((B, 2), -1, dtype=torch.int32)
    g = torch.Generator().manual_seed(1)
    for b in range(B):
        p = torch.randperm(4, generator=g)
        sel[b, 0] = p[0]
        if b < pairs - B: sel[b, 1] = p[1]
    sel = sel.to(dev)
    gate = torch.rand(B, E, device=dev) * 0.5 + 0.25
    y = torch.empty(B, HW, HW, Cc, device=dev, dtype=bf)
    s = torch.cuda.current_stream().cuda_stream
    line = f"C{Cc} @{HW}x{HW} pairs {int((sel >= 0).sum())}: "
    for v in sys.argv[1:]:
        lib = C.CDLL(f"/tmp/esfab/libesf_{v}.so")
        call = lambda: lib.ymk_esmoe_fused(1, vp(x), B, HW, HW, Cc, Cc, vp(dw_w), vp(dw_off), vp(ksz), 9, Cc, Cc, vp(pw_w), vp(pw_b), vp(ns), vp(nt), E, top_k, vp(sel), vp(gate), vp(y), Cc, C.c_void_p(s))
        for _ in range(3): assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        line += f" [{v}] {e0.elapsed_time(e1) * 100:.0f}us"
    print(line)
PY
