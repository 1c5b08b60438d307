"""Depthwise kernel at equal bytes and tiles but different channel interleave (B x C = 64 x 128 ... 512 x 16), GPU box; needs
tools/micro/_dwab/libdw_0.so (tools/micro/dwv_ablate.sh build).  Result of round 2: profiles/r02_dw_ablation_valu.txt."""
import ctypes as C, sys, torch
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
vp = lambda t: C.c_void_p(t.data_ptr())
lib = C.CDLL("tools/micro/_dwab/libdw_0.so")
print("same bytes and tiles, different channel interleave (us per call, k = 3 5 7 9)")
for B, H, Cc in ((64, 160, 128), (128, 160, 64), (256, 160, 32), (512, 160, 16), (64, 160, 128)):
    x = torch.randn(B, H, H, Cc, generator=g).to(bf).cuda()
    out = torch.empty_like(x)
    row = []
    for k in (3, 5, 7, 9):
        w = (torch.randn(k * k, Cc, generator=g) / k).to(bf).cuda()
        s = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.ymk_dwconv2d(1, vp(x), vp(w), None, None, vp(out), B, H, H, Cc, k, Cc, Cc, 0, 0, C.c_void_p(s))
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 100)
    print(f"B {B:4d} C {Cc:4d}" + "".join(f"{t:10.1f}" for t in row))
