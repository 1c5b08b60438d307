#!/bin/bash
# GPU box: depthwise stencil (csrc/dwconv.hip) with all LDS reads of a filter row issued before the first use (default) vs the compiler's
# own read-wait-use schedule (-DDW_SERIAL_READS), on the S detector's depthwise shapes at batch 64.
cd "$(dirname "$0")/../.."
mkdir -p /tmp/dwab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared yolo_master_amd/csrc/dwconv.hip -o /tmp/dwab/libdw_batch.so 2>/dev/null &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDW_SERIAL_READS yolo_master_amd/csrc/dwconv.hip -o /tmp/dwab/libdw_serial.so 2>/dev/null &
wait
python - <<'PY'
import ctypes as C, torch
dev, bf = "cuda", torch.bfloat16
def vp(t): return C.c_void_p(t.data_ptr())
def timeit(call):
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100
libs = {k: C.CDLL(f"/tmp/dwab/libdw_{k}.so") for k in ("serial", "batch")}
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for Cc, HW, pairs in [(128, 160, 83), (256, 80, 77), (256, 40, 117), (512, 20, 84)]:
    B, E, top_k, ks = 64, 4, 2, [3, 5, 7, 9]
    x = torch.randn(B, HW, HW, Cc, device=dev).to(bf)
    parts, offs, off = [], [], 0
    for k in ks:
        w = (torch.randn(k * k, Cc, device=dev) / k).to(bf); parts.append(w.reshape(-1)); offs.append(off); off += w.numel()
    dw_w = torch.cat(parts); dw_off = torch.tensor(offs, dtype=torch.int32, device=dev); ksz = torch.tensor(ks, dtype=torch.int32, device=dev)
    sel = torch.full((B, 2), -1, dtype=torch.int32)
    g = torch.Generator().manual_seed(1)
    for b in range(B):
        p = torch.randperm(4, generator=g); sel[b, 0] = p[0]
        if b < pairs - B: sel[b, 1] = p[1]
    sel = sel.to(dev)
    out = torch.empty(B * 2, HW, HW, Cc, device=dev, dtype=bf)
    dummy = torch.zeros(8, dtype=torch.int32, device=dev)
    line = f"moe_dw C{Cc} @{HW}x{HW} pairs {int((sel >= 0).sum())}:"
    for k, lib in libs.items():
        t = timeit(lambda: lib.ymk_esmoe_dw(1, vp(x), B, HW, HW, Cc, Cc, vp(dw_w), vp(dw_off), vp(ksz), E, top_k, 9, vp(sel), vp(dummy), vp(dummy), vp(out), s))
        line += f"  {k} {t:.0f} us"
    print(line)
for Cc, HW, k in [(128, 40, 7), (256, 20, 7), (128, 80, 3), (128, 40, 3), (128, 20, 3)]:
    B = 64
    x = torch.randn(B, HW, HW, Cc, device=dev).to(bf); w = torch.randn(k * k, Cc, device=dev).to(bf); y = torch.empty_like(x)
    line = f"dwconv C{Cc} k{k} @{HW}x{HW}:"
    for kk, lib in libs.items():
        t = timeit(lambda: lib.ymk_dwconv2d(1, vp(x), vp(w), None, None, vp(y), B, HW, HW, Cc, k, Cc, Cc, 0, 0, s))
        line += f"  {kk} {t:.0f} us"
    print(line)
PY
