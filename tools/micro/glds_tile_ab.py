"""Two builds of csrc/conv_glds.hip side by side on the 3x3 / 1x1 shapes of the S detector (GPU box).  Usage:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Iyolo_master_amd/csrc [-D...] yolo_master_amd/csrc/conv_glds.hip -o tools/micro/_dwab/libglds_old.so
  (the same for libglds_new.so from the other source / flags; build in the container, the directory is git-ignored and travels with gpurun)
  python tools/micro/glds_tile_ab.py        # us per call, two- and three-stage loop of each build; flags outputs that are not bit-identical
Results of round 2: profiles/r02_glds_tile_ab.txt."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from yolo_master_amd import _lib, ops
bf = torch.bfloat16
p = lambda t: C.c_void_p(t.data_ptr())
shapes = [(64, 64, 3, 1, 20), (256, 64, 3, 1, 40), (128, 64, 3, 1, 80), (128, 128, 3, 2, 80), (256, 256, 3, 2, 40), (256, 512, 3, 2, 40), (256, 256, 3, 2, 80), (128, 128, 3, 2, 160), (384, 256, 1, 1, 40), (256, 256, 1, 1, 20), (256, 768, 1, 1, 20), (768, 512, 1, 1, 20), (256, 128, 1, 1, 40)]
st = torch.cuda.current_stream().cuda_stream
libs = {n: C.CDLL(f"tools/micro/_dwab/{n}.so") for n in ("libglds_old", "libglds_new")}
print(f"{'shape':28s} {'old s2':>9s} {'old s3':>9s} {'new s2':>9s} {'new s3':>9s}")
for cin, cout, k, s, hw in shapes:
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(64, hw, hw, cin, generator=g).to(bf).cuda()
    w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, generator=g) * (k * k * cin) ** -0.5, bf).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    ho = (hw + 2 * (k // 2) - k) // s + 1
    ys = {}
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, 64, hw, hw, cin, cout, k, s, cin, cout, 0, w.shape[1], _lib.ACT_SILU)
    row = []
    for n, lib in libs.items():
        for two in (1, 0):
            y = torch.empty((64, ho, ho, cout), dtype=bf, device="cuda")
            call = lambda: lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), two, C.c_void_p(st))
            for _ in range(3): call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): rc = call()
            e1.record(); torch.cuda.synchronize()
            assert rc == 0
            row.append(e0.elapsed_time(e1) * 50)
            ys[(n, two)] = y
    ref = ys[("libglds_old", 1)].float()
    bad = [k2 for k2, v in ys.items() if not torch.equal(v.float(), ref)]
    print(f"{cin:4d}->{cout:<4d} k{k} s{s} in {hw:3d}^2    " + "".join(f"{t:10.1f}" for t in row) + ("   MISMATCH " + str(bad) if bad else ""))
