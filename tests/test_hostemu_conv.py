"""The opt-in tiled convolution core (csrc/conv_glds.hip, include/ymk_next.h) on the CPU lane emulator (tests/hostemu:
matrix core, LDS-DMA, barriers and the XCD tile order emulated; DMA completes at once): the library entry point
`ymk_conv2d_glds`, called through its ctypes binding with the library's own descriptor, against the torch restatement of
the `ymk_conv2d` contract — strided input / output / residual views, both activation codes, bf16 and fp32 outputs,
1x1 and 3x3, stride 1 and 2, both cout tile widths, tail tiles, both k-loop flavours."""
import ctypes as C

import pytest
import torch

from tests import emu_ops


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _wide(t, pad, dev="cpu"):
    """Same values as a channel slice of a wider buffer on `dev` (pad = extra channels of the buffer)."""
    if not pad:
        return t.contiguous().to(dev)
    buf = torch.zeros((*t.shape[:3], t.shape[3] + pad), dtype=t.dtype)
    buf[..., : t.shape[3]] = t
    return buf.to(dev)[..., : t.shape[3]]


CASES = [
    # B, H, W, Cin, Cout, k, s, act, residual, out fp32, x pad, y pad, two_stage
    (2, 13, 11, 64, 128, 3, 1, True, True, False, 0, 0, 0),
    (2, 23, 19, 64, 128, 3, 2, True, False, False, 64, 0, 1),
    (1, 18, 17, 128, 64, 3, 1, False, True, False, 0, 64, 0),
    (3, 9, 11, 192, 192, 1, 1, True, False, True, 0, 0, 0),      # BN = 64, three cout tiles, fp32 logits-style output
    (1, 40, 13, 64, 256, 3, 1, True, False, False, 0, 128, 1),   # two cout tiles per pixel tile, three pixel tiles
    (2, 16, 16, 128, 128, 1, 2, False, False, False, 8, 4, 0),   # strided 1x1
    (1, 3, 5, 64, 64, 3, 1, True, True, True, 0, 0, 0),          # a single ragged tile
]


def run_case(lib, case, dev="cpu", stream=None):
    """One parity case of ymk_conv2d_glds against the contract restatement; shared with tests/test_gpu_next.py."""
    from yolo_master_amd import _lib, ops

    B, H, W, Cin, Cout, k, s, act, use_res, out_f32, xpad, ypad, two = case
    bf = torch.bfloat16
    x = _rnd(B, H, W, Cin, seed=1).to(bf)
    wp = ops.pack_conv_weight(_rnd(Cout, Cin, k, k, seed=2, scale=(k * k * Cin) ** -0.5), bf)
    bias = _rnd(Cout, seed=3, scale=0.2)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    res = _rnd(B, Ho, Wo, Cout, seed=4).to(bf) if use_res else None
    odt = torch.float32 if out_f32 else bf
    ref = emu_ops.conv2d(x, wp, bias, k, s, act, residual=res, out_dtype=odt)
    ybuf = torch.full((B, Ho, Wo, Cout + ypad), 7.0, dtype=odt, device=dev)
    y = ybuf[..., :Cout]
    xd, rd = _wide(x, xpad, dev), (None if res is None else _wide(res, 32, dev))
    wd, bd = wp.to(dev), bias.to(dev)
    d = _lib.ConvDesc(_lib.YMK_BF16, ops.DT[odt], B, H, W, Cin, Cout, k, s, xd.stride(2), y.stride(2), rd.stride(2) if use_res else 0,
                      wp.shape[1], {"gelu": _lib.ACT_GELU, "sigmoid": _lib.ACT_SIGMOID}.get(act, _lib.ACT_SILU if act else _lib.ACT_NONE))
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())   # noqa: E731
    entry = lib.ymk_conv2d_glds if two is not None else None
    rc = entry(C.byref(d), p(xd), p(wd), p(bd), p(rd), p(y), two, stream) if entry else lib.ymk_conv2d(C.byref(d), p(xd), p(wd), p(bd), p(rd), p(y), stream)
    assert rc == 0
    got = y.float().cpu()
    err = float((got - ref.float()).abs().max())
    tol = (2e-5 if out_f32 else 1.6e-2) * max(1.0, float(ref.float().abs().max()))
    assert err <= tol, f"max |d| {err:.3e}"
    if ypad:
        assert float((ybuf[..., Cout:].float().cpu() - 7.0).abs().max()) == 0.0, "wrote outside its channel slice"


@pytest.mark.parametrize("case", CASES)
def test_conv2d_glds_on_the_emulator(hostlib, case):
    run_case(hostlib, case)


EPILOGUE_CASES = [
    # the extended epilogues (ymk.h YMK_ACT_GELU / YMK_ACT_SIGMOID; the token FFNs of the MoT experts, the gates of the gated MoE):
    # in the LDS-DMA core's epilogue ...
    (2, 9, 11, 64, 128, 1, 1, "gelu", False, False, 0, 0, 1), (1, 13, 7, 128, 64, 3, 1, "sigmoid", False, True, 64, 8, 1),
    # ... and through ymk_conv2d (two = None): a shape the core takes, and shapes that go to the other cores (one in-place pass follows)
    (1, 20, 20, 128, 256, 1, 1, "gelu", False, False, 0, 0, None), (2, 7, 9, 32, 48, 1, 1, "gelu", False, False, 0, 8, None),
    (1, 6, 5, 16, 24, 3, 1, "sigmoid", False, True, 0, 0, None),
]


@pytest.mark.parametrize("case", EPILOGUE_CASES)
def test_conv2d_gelu_and_sigmoid_epilogues(hostlib, case):
    run_case(hostlib, case)


def test_conv2d_extended_epilogues_take_no_residual(hostlib):
    from yolo_master_amd import _lib, ops

    x = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16)
    w = ops.pack_conv_weight(torch.zeros(64, 64, 1, 1), torch.bfloat16)
    b, y = torch.zeros(64), torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16)
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, 1, 8, 8, 64, 64, 1, 1, 64, 64, 64, 64, _lib.ACT_GELU)
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    assert hostlib.ymk_conv2d(C.byref(d), p(x), p(w), p(b), p(y), p(y), None) == -1
    assert hostlib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(b), p(y), p(y), 1, None) == -1


def tile_flags(bn, bm, two=1):
    """`two_stage` argument of the LDS-DMA entry points with a forced tile shape (csrc/conv_glds.hip glds_launch_any)."""
    return two | ((bn // 64) << 8) | (bm << 12)


BIG_TILE_CASES = [
    # the round-3 tiles (one workgroup per CU, 160 / 128 KB of LDS): 128 couts x 512 pixels and 256 x 256, ragged in pixels, with
    # residual (loaded per fragment there), stride 2, channel-slice views, fp32 output
    ((1, 24, 23, 64, 128, 3, 1, True, True, False, 0, 64, 1), (128, 512)),
    ((2, 21, 19, 64, 256, 3, 2, True, False, False, 64, 0, 1), (256, 256)),
    ((1, 18, 17, 128, 256, 1, 1, False, True, True, 0, 0, 1), (256, 256)),
    ((1, 30, 20, 64, 128, 1, 1, True, False, False, 0, 0, 1), (128, 512)),
    ((1, 12, 12, 64, 128, 3, 1, True, True, False, 0, 0, 1), (128, 256)),
    # round 6: 256 couts x 208 pixels (13 pixel fragments as 7 + 6 over the two wave rows; the last transfer instruction of a k-step only in the
    # waves whose rows exist): two tiles + a ragged third, stride 2 with residual; 1x1 with fp32 output; two cout tiles
    ((2, 21, 19, 64, 256, 3, 2, True, True, False, 64, 0, 1), (256, 208)),
    ((1, 25, 18, 128, 256, 1, 1, False, False, True, 0, 0, 1), (256, 208)),
    ((1, 16, 13, 64, 512, 3, 1, True, False, False, 0, 128, 1), (256, 208)),
]


@pytest.mark.parametrize("case,tile", BIG_TILE_CASES)
def test_conv2d_glds_big_tiles_on_the_emulator(hostlib, case, tile):
    run_case(hostlib, case[:-1] + (tile_flags(*tile),))


def test_conv2d_glds_rejects_impossible_tiles(hostlib):
    from yolo_master_amd import _lib, ops

    x = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16)
    w = ops.pack_conv_weight(torch.zeros(128, 64, 1, 1), torch.bfloat16)
    b, y = torch.zeros(128), torch.zeros(1, 8, 8, 128, dtype=torch.bfloat16)
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, 1, 8, 8, 64, 128, 1, 1, 64, 128, 0, 64, 0)
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    for bn, bm, two in ((256, 512, 1), (128, 512, 0), (192, 128, 1), (256, 128, 1), (128, 208, 1), (256, 208, 0)):   # 3-stage big tiles do not fit the LDS
        assert hostlib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(b), None, p(y), tile_flags(bn, bm, two), None) == -1


def test_conv2d_glds_rejects_what_it_does_not_cover(hostlib):
    from yolo_master_amd import _lib

    x = torch.zeros((1, 4, 4, 64), dtype=torch.bfloat16)
    w = torch.zeros((64, 576), dtype=torch.bfloat16)
    b = torch.zeros(64)
    y = torch.zeros((1, 4, 4, 64), dtype=torch.bfloat16)
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    for desc in (_lib.ConvDesc(_lib.YMK_F32, _lib.YMK_F32, 1, 4, 4, 64, 64, 3, 1, 64, 64, 0, 576, 1),        # fp32 compute
                 _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, 1, 4, 4, 32, 64, 3, 1, 32, 64, 0, 320, 1),      # Cin % 64
                 _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, 1, 4, 4, 64, 80, 3, 1, 64, 80, 0, 576, 1),      # Cout % 64
                 _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, 1, 4, 4, 64, 64, 3, 1, 64, 64, 0, 640, 1)):     # padded K
        assert hostlib.ymk_conv2d_glds(C.byref(desc), p(x), p(w), p(b), None, p(y), 0, None) == -1


CAT2_CASES = [
    # B, H, W, C1, C2, Cout, upsample first source, act, x1 pad, x2 pad, y pad, two_stage
    (2, 10, 14, 64, 64, 128, True, True, 0, 0, 0, 0),
    (1, 12, 10, 128, 64, 64, False, True, 64, 8, 64, 1),
    (2, 6, 22, 64, 192, 192, True, False, 0, 0, 0, 0),        # BN = 64, ragged tail, three cout tiles
    (3, 8, 8, 256, 128, 128, False, True, 0, 64, 0, 0),
    (2, 10, 14, 64, 192, 256, True, True, 0, 0, 0, 1 | (4 << 8) | (256 << 12)),     # 256 x 256 tile, the source switches at k-step 1
    (1, 12, 10, 128, 64, 128, False, True, 64, 8, 64, 1 | (2 << 8) | (256 << 12)),
]


def run_cat2_case(lib, case, dev="cpu", stream=None):
    from yolo_master_amd import _lib, ops

    B, H, W, C1, C2, Cout, up, act, p1, p2, py, two = case
    bf = torch.bfloat16
    h1, w1 = (H // 2, W // 2) if up else (H, W)
    x1, x2 = _rnd(B, h1, w1, C1, seed=11).to(bf), _rnd(B, H, W, C2, seed=12).to(bf)
    wp = ops.pack_conv_weight(_rnd(Cout, C1 + C2, 1, 1, seed=13, scale=(C1 + C2) ** -0.5), bf)
    bias = _rnd(Cout, seed=14, scale=0.2)
    ref = emu_ops.conv1x1_cat2(x1, up, x2, wp, bias, act)
    ybuf = torch.full((B, H, W, Cout + py), 7.0, dtype=bf, device=dev)
    y = ybuf[..., :Cout]
    x1d, x2d, wd, bd = _wide(x1, p1, dev), _wide(x2, p2, dev), wp.to(dev), bias.to(dev)
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, B, H, W, C1 + C2, Cout, 1, 1, x1d.stride(2), y.stride(2), 0, wp.shape[1],
                      _lib.ACT_SILU if act else _lib.ACT_NONE)
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    rc = lib.ymk_conv1x1_cat2_glds(C.byref(d), p(x1d), C1, x1d.stride(2), int(up), p(x2d), x2d.stride(2), p(wd), p(bd), p(y), two, stream)
    assert rc == 0
    err = float((y.float().cpu() - ref.float()).abs().max())
    assert err <= 1.6e-2 * max(1.0, float(ref.float().abs().max())), f"max |d| {err:.3e}"
    if py:
        assert float((ybuf[..., Cout:].float().cpu() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("case", CAT2_CASES)
def test_conv1x1_cat2_glds_on_the_emulator(hostlib, case):
    run_cat2_case(hostlib, case)


EXPERT_CASES = [
    # B, H, W, Cin, Cout, k, E, idx, two_stage
    (2, 9, 7, 64, 64, 3, 4, [[2, 0], [1, 3]], 0),
    (3, 17, 16, 128, 128, 3, 4, [[0, 1], [3, 3], [2, 0]], 1),         # image larger than one tile (272 pixels): tail tile per image
    (2, 6, 5, 192, 64, 1, 16, [[15, 4, 9], [0, 15, 7]], 0),           # shared-inverted projections: 1x1, sixteen banks, three slots
    (2, 17, 16, 64, 128, 3, 4, [[0, 1], [3, 2]], 1 | (2 << 8) | (256 << 12)),   # 128 x 256 tiles, two per image
]


def run_expert_case(lib, case, dev="cpu", stream=None):
    from yolo_master_amd import _lib

    B, H, W, Cin, Cout, k, E, idx, two = case
    bf = torch.bfloat16
    x = _rnd(B, H, W, Cin, seed=21).to(bf)
    Kp = k * k * Cin
    wp = (_rnd(E, Cout, Kp, seed=22) * Kp ** -0.5).to(bf)
    it = torch.tensor(idx, dtype=torch.int32)
    ref = emu_ops.expert_conv(x, wp, k, it)
    K = it.shape[1]
    out = torch.full((K * B, H, W, Cout), 7.0, dtype=bf, device=dev)
    xd, wd, idd = x.to(dev), wp.to(dev), it.to(dev)
    d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, B, H, W, Cin, Cout, k, 1, Cin, Cout, 0, Kp, _lib.ACT_NONE)
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    rc = lib.ymk_expert_conv_glds(C.byref(d), p(xd), p(wd), p(idd), K, E, p(out), two, stream)
    assert rc == 0
    err = float((out.float().cpu() - ref.float()).abs().max())
    assert err <= 1.6e-2 * max(1.0, float(ref.float().abs().max())), f"max |d| {err:.3e}"


@pytest.mark.parametrize("case", EXPERT_CASES)
def test_expert_conv_glds_on_the_emulator(hostlib, case):
    run_expert_case(hostlib, case)


@pytest.mark.parametrize("seed", range(16))
def test_conv2d_glds_random_shapes(hostlib, seed):
    """Seeded sweep over kernel size, stride, channel counts, map sizes (single pixels to several tiles), views, epilogue
    options and both k-loops."""
    import random

    rng = random.Random(500 + seed)
    k = rng.choice([1, 3])
    case = (rng.randint(1, 3), rng.randint(1, 21), rng.randint(1, 21), rng.choice([64, 128, 192]), rng.choice([64, 128, 192, 256]), k,
            rng.choice([1, 2]), rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.3, rng.choice([0, 8, 64]), rng.choice([0, 4, 64]),
            rng.choice([0, 1]))
    run_case(hostlib, case)
    if seed % 4 == 0:   # and the virtual-concatenation / routed-expert forms on shapes drawn the same way
        H, W = 2 * rng.randint(1, 8), 2 * rng.randint(1, 8)
        run_cat2_case(hostlib, (rng.randint(1, 2), H, W, rng.choice([64, 128]), rng.choice([64, 192]), rng.choice([64, 128]), rng.random() < 0.5,
                                rng.random() < 0.5, rng.choice([0, 8]), rng.choice([0, 64]), rng.choice([0, 64]), rng.choice([0, 1])))
        B, E, K = rng.randint(1, 3), rng.choice([2, 5]), rng.randint(1, 2)
        idx = [[rng.randrange(E) for _ in range(K)] for _ in range(B)]
        run_expert_case(hostlib, (B, rng.randint(1, 18), rng.randint(1, 18), rng.choice([64, 128]), rng.choice([64, 128]), rng.choice([1, 3]), E, idx,
                                  rng.choice([0, 1])))
