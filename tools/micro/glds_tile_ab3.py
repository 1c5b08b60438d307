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
# (couts, pixels, 0, 0) — the last two were the half-k-step / out-of-phase variants of round 3 (tools/micro/parked/conv_glds_pp_k32.hip.txt)
tiles = [(64, 128, 0, 0), (64, 256, 0, 0), (128, 128, 0, 0), (128, 256, 0, 0), (128, 512, 0, 0), (256, 256, 0, 0)]
st = torch.cuda.current_stream().cuda_stream
print(f"{'shape (batch %d)' % B:30s} " + "".join(f"{f'{bn}x{bm}' + ('h' if k32 else '') + ('p' if pp else ''):>10s}" for bn, bm, k32, pp in tiles) + f"{'default':>10s}   GFLOP   best TF/s")
for cin, cout, k, s, hw in shapes:
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(B, hw, hw, cin, generator=g).to(bf).cuda()
    w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, generator=g) * (k * k * cin) ** -0.5, bf).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    ho = (hw + 2 * (k // 2) - k) // s + 1
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, B, hw, hw, cin, cout, k, s, cin, cout, 0, w.shape[1], _lib.ACT_SILU)
    # the power state ramps over tens of milliseconds and a column measured later in a row was measured faster (the default rule, timed last,
    # beat the identical forced tile by 10 %): warm the clocks first, then interleave the variants round-robin and take each one's median
    variants, ref, bad = [], None, []
    for bn, bm, k32, pp in tiles + [(0, 0, 0, 0)]:
        if bn and cout % bn:
            variants.append(None)
            continue
        flags = 1 | ((bn // 64) << 8) | (bm << 12) | (k32 << 22) | (pp << 23)
        y = torch.empty((B, ho, ho, cout), dtype=bf, device="cuda")
        rc = lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), flags, C.c_void_p(st))
        assert rc == 0, (cin, cout, k, s, hw, bn, bm, rc)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.float()
        elif not torch.equal(y.float(), ref):
            bad.append((bn, bm, k32, pp))
        variants.append((flags, y))
    live = [v for v in variants if v]
    for _ in range(6):
        for flags, y in live:
            for _ in range(5):
                lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), flags, C.c_void_p(st))
    torch.cuda.synchronize()
    samples = {id(v): [] for v in live}
    for _ in range(7):
        for v in live:
            flags, y = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), flags, C.c_void_p(st))
            e1.record()
            torch.cuda.synchronize()
            samples[id(v)].append(e0.elapsed_time(e1) * 125)
    row = [float("nan") if v is None else sorted(samples[id(v)])[3] for v in variants]
    gf = 2.0 * B * ho * ho * cout * cin * k * k / 1e9
    best = min(t for t in row if t == t)
    print(f"{cin:4d}->{cout:<4d} k{k} s{s} in {hw:3d}^2      " + "".join(f"{t:10.1f}" for t in row) + f"   {gf:6.1f}   {gf / best * 1e-3 * 1e3:8.0f}" +
          ("   MISMATCH " + str(bad) if bad else ""))
