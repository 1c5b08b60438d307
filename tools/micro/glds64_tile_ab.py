"""64-cout shapes of the LDS-DMA core: 128- against 256-pixel tiles per shape, interleaved in one process (the rule in glds_launch_any picks
128 for launches of >= 1024 tiles since round 2).  python tools/micro/glds64_tile_ab.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_master_amd import _lib, ops  # noqa: E402

lib = _lib.load()
bf = torch.bfloat16
p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
SHAPES = [(128, 64, 3, 1, 80), (256, 64, 3, 1, 40), (256, 64, 3, 1, 20), (64, 64, 3, 1, 40), (64, 64, 3, 1, 20), (64, 64, 1, 1, 80), (64, 64, 1, 1, 40)]
st = torch.cuda.current_stream().cuda_stream
print(f"{'shape':24s} {'64x128 us':>10s} {'64x256 us':>10s} {'rule us':>9s}")
for cin, cout, k, s, hw in SHAPES:
    g = torch.Generator().manual_seed(cin + k)
    x = torch.randn(64, hw, hw, cin, generator=g).to(bf).cuda()
    w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, generator=g) * (k * k * cin) ** -0.5, bf).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    y = torch.empty((64, hw, hw, cout), dtype=bf, device="cuda")
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, 64, hw, hw, cin, cout, k, s, cin, cout, 0, w.shape[1], _lib.ACT_SILU)
    res = []
    for flags in (1 | (1 << 8) | (128 << 12), 1 | (1 << 8) | (256 << 12), 1):
        best = 1e9
        for rep in range(3):
            for _ in range(20):
                rc = lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), flags, C.c_void_p(st))
            assert rc == 0, rc
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), flags, C.c_void_p(st))
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 50)
        res.append(best)
    print(f"{'%d->%d k%d s%d @%d' % (cin, cout, k, s, hw):24s} {res[0]:10.1f} {res[1]:10.1f} {res[2]:9.1f}")
