"""Tile-shape A/B of the LDS-DMA convolution core (csrc/conv_glds.hip) on the shapes the S detector runs at 64 images (GPU box):
us per call for each tile the `two_stage` flags can force (bits 8-11 cout-tile / 64, bits 12-21 pixel-tile), bit-identity of the outputs,
and what the default rule picks.  Round 3: profiles/r03_glds_tile_ab.txt.    python tools/micro/glds_tile_ab3.py [batch=64]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from yolo_master_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bf = torch.bfloat16
p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
lib = _lib.load()
# cin, cout, k, s, input hw
shapes = [(128, 128, 3, 2, 160), (256, 256, 3, 2, 80), (256, 512, 3, 2, 40), (128, 128, 3, 2, 80), (256, 256, 3, 2, 40),
          (128, 64, 3, 1, 80), (256, 64, 3, 1, 40), (64, 64, 3, 1, 40), (64, 64, 3, 1, 20),
          (512, 128, 1, 1, 80), (768, 256, 1, 1, 40), (384, 256, 1, 1, 40), (768, 512, 1, 1, 20), (256, 768, 1, 1, 20), (512, 256, 1, 1, 20),
          (256, 256, 1, 1, 20), (256, 128, 1, 1, 40), (128, 64, 1, 1, 40), (64, 64, 1, 1, 80)]
# (couts, pixels, half k-steps, out-of-phase loop)
tiles = [(64, 128, 0, 0), (128, 128, 0, 0), (256, 256, 0, 0), (128, 256, 1, 0), (128, 128, 0, 1), (128, 256, 0, 1), (256, 256, 0, 1), (128, 256, 1, 1), (256, 256, 1, 1)]
st = torch.cuda.current_stream().cuda_stream
print(f"{'shape (batch %d)' % B:30s} " + "".join(f"{f'{bn}x{bm}' + ('h' if k32 else '') + ('p' if pp else ''):>10s}" for bn, bm, k32, pp in tiles) + f"{'default':>10s}   GFLOP   best TF/s")
for cin, cout, k, s, hw in shapes:
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(B, hw, hw, cin, generator=g).to(bf).cuda()
    w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, generator=g) * (k * k * cin) ** -0.5, bf).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    ho = (hw + 2 * (k // 2) - k) // s + 1
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, B, hw, hw, cin, cout, k, s, cin, cout, 0, w.shape[1], _lib.ACT_SILU)
    row, ref, bad = [], None, []
    for bn, bm, k32, pp in tiles + [(0, 0, 0, 0)]:
        if bn and cout % bn:
            row.append(float("nan"))
            continue
        flags = 1 | ((bn // 64) << 8) | (bm << 12) | (k32 << 22) | (pp << 23)
        y = torch.empty((B, ho, ho, cout), dtype=bf, device="cuda")
        call = lambda: lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), flags, C.c_void_p(st))   # noqa: E731
        for _ in range(3):
            rc = call()
        assert rc == 0, (cin, cout, k, s, hw, bn, bm, rc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 50)
        if ref is None:
            ref = y.float()
        elif not torch.equal(y.float(), ref):
            bad.append((bn, bm, k32, pp))
    gf = 2.0 * B * ho * ho * cout * cin * k * k / 1e9
    best = min(t for t in row if t == t)
    print(f"{cin:4d}->{cout:<4d} k{k} s{s} in {hw:3d}^2      " + "".join(f"{t:10.1f}" for t in row) + f"   {gf:6.1f}   {gf / best * 1e-3 * 1e3:8.0f}" +
          ("   MISMATCH " + str(bad) if bad else ""))
