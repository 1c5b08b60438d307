"""csrc/conv.hip and csrc/dwconv.hip on the CPU lane emulator (tests/hostemu) through the product's wrappers: the streaming 1x1
kernel with its weight-row permutation for 16-byte stores (whole 64-cout groups) and without it, the spatial-tile 3x3 kernel, the
tiled implicit GEMM, and the depthwise stencil on both tile shapes.  References: the torch restatements of the C-ABI contracts
(tests/emu_ops.py).  Logic only — routing thresholds are lowered for the emulator (tests/conftest.py: YMK_WS_MIN_TILES)."""
import pytest
import torch

from tests import emu_ops


@pytest.fixture
def host_ops(hostlib, monkeypatch):
    from yolo_master_amd import ops

    monkeypatch.setattr(ops, "lib", hostlib)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    return ops


def _rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


CONV_CASES = [
    # dtype, B, H, W, Cin, Cout, k, stride, act, residual, output pitch pad, expected kernel family (ymk_conv2d_last_variant & 0xff)
    (torch.bfloat16, 2, 16, 16, 128, 128, 1, 1, True, True, 0, 1),     # streaming 1x1, permuted rows, 16-byte stores
    (torch.bfloat16, 2, 16, 16, 128, 128, 1, 1, True, False, 4, 1),    # ... row pitch not a multiple of 8: the 8-byte store pair
    (torch.bfloat16, 1, 16, 24, 96, 192, 1, 1, False, True, 0, 1),     # two cout tiles, the second half empty; K padded to 128
    (torch.bfloat16, 1, 16, 16, 64, 80, 1, 1, True, False, 0, 1),      # Cout not a multiple of 64: natural row order
    (torch.float32, 1, 16, 16, 64, 128, 1, 1, True, True, 0, 1),       # fp32 streaming 1x1 (two K groups of 32)
    (torch.bfloat16, 8, 64, 64, 32, 32, 3, 1, True, True, 0, 2),       # spatial-tile 3x3 (128 tiles), residual prefetch
    (torch.bfloat16, 8, 64, 64, 32, 64, 3, 1, True, False, 0, 2),      # ... four cout row blocks
    (torch.bfloat16, 8, 64, 64, 16, 32, 3, 1, False, True, 4, 2),      # ... paired taps (Cin 16), padded output pitch
    (torch.float32, 2, 9, 11, 16, 24, 3, 2, True, False, 0, 0),        # tiled implicit GEMM, strided, ragged
    (torch.bfloat16, 2, 7, 9, 64, 64, 1, 1, False, True, 0, 0),        # tiled 1x1 below every threshold
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"{str(c[0])[6:]}-{c[4]}to{c[5]}-k{c[6]}s{c[7]}-{c[2]}x{c[3]}")
def test_conv2d_kernels(case, host_ops, hostlib):
    dtype, B, H, W, Cin, Cout, k, s, act, use_res, ypad, family = case
    x = _rnd(B, H, W, Cin, seed=1).to(dtype)
    wp = host_ops.pack_conv_weight(_rnd(Cout, Cin, k, k, seed=2, scale=(k * k * Cin) ** -0.5), dtype)
    bias = _rnd(Cout, seed=3, scale=0.2)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    res = _rnd(B, Ho, Wo, Cout, seed=4).to(dtype) if use_res else None
    ref = emu_ops.conv2d(x, wp, bias, k, s, act, residual=res)
    ybuf = torch.full((B, Ho, Wo, Cout + ypad), 7.0, dtype=dtype)
    out = ybuf[..., :Cout]
    got = host_ops.conv2d(x, wp, bias, k, s, act, out=out, residual=res)
    assert (hostlib.ymk_conv2d_last_variant() & 0xff) == family, "the shape did not reach the kernel family under test"
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert torch.allclose(got.float(), ref.float(), atol=tol, rtol=tol), float((got.float() - ref.float()).abs().max())
    if ypad:
        assert torch.all(ybuf[..., Cout:] == 7.0), "wrote past the channel range"


POOL_CASES = [
    # B, H, W, Cin, Cout, act, residual
    (2, 16, 16, 128, 128, True, False),     # two 128-pixel tiles per image, one cout tile
    (3, 16, 24, 96, 192, True, True),       # three tiles per image, two cout tiles (the second half empty), K padded to 128, residual
    (1, 32, 16, 192, 256, False, False),    # three K groups, two full cout tiles, no activation
]


@pytest.mark.parametrize("case", POOL_CASES, ids=lambda c: f"{c[3]}to{c[4]}-{c[0]}x{c[1]}x{c[2]}")
def test_streaming_1x1_leaves_the_routers_pooled_sums(case, host_ops, hostlib):
    """ops.conv2d(pool=True) -> ymk_conv1x1_pooled: the same output as the plain convolution, bit for bit, plus `out.gap_part` = the sums
    of the STORED values over the 128-pixel tiles of each image (what ymk_esmoe_route_pooled consumes instead of the map)."""
    B, H, W, Cin, Cout, act, use_res = case
    dtype = torch.bfloat16
    x = _rnd(B, H, W, Cin, seed=11).to(dtype)
    wp = host_ops.pack_conv_weight(_rnd(Cout, Cin, 1, 1, seed=12, scale=Cin ** -0.5), dtype)
    bias = _rnd(Cout, seed=13, scale=0.2)
    res = _rnd(B, H, W, Cout, seed=14).to(dtype) if use_res else None
    plain = host_ops.conv2d(x, wp, bias, 1, 1, act, residual=res)
    assert getattr(plain, "gap_part", None) is None
    got = host_ops.conv2d(x, wp, bias, 1, 1, act, residual=res, pool=True)
    assert (hostlib.ymk_conv2d_last_variant() & 0xff) == 1
    assert torch.equal(got, plain), "the pooled variant stores other values"
    part = got.gap_part
    chunks = H * W // 128
    assert part.shape == (B, chunks, Cout) and part.dtype == torch.float32
    ref = got.float().reshape(B, chunks, 128, Cout).sum(2)
    assert torch.allclose(part, ref, rtol=1e-5, atol=1e-4), float((part - ref).abs().max())
    # the same sums, bit for bit, from the stand-alone kernel (what a batch too small for the streaming kernel gets)
    import ctypes as C
    alone = torch.empty_like(part)
    assert hostlib.ymk_pool_tiles128(1, C.c_void_p(got.data_ptr()), Cout, B, H * W, Cout, C.c_void_p(alone.data_ptr()), None) == 0
    assert torch.equal(alone, part), float((alone - part).abs().max())
    # a map that is not a whole number of 128-pixel tiles gets no sums (the router reads it)
    odd = host_ops.conv2d(x[:, :, :15].contiguous(), wp, bias, 1, 1, act, pool=True)   # H x 15 pixels
    assert getattr(odd, "gap_part", None) is None


DW_CASES = [
    # dtype, B, H, W, C, k, bias, act, residual
    (torch.bfloat16, 2, 20, 23, 32, 3, True, True, False),     # 40-wide tiles, two channel blocks
    (torch.bfloat16, 1, 33, 41, 16, 9, False, False, True),    # more than one tile each way, the largest ES-MoE stencil
    (torch.bfloat16, 2, 17, 18, 24, 7, True, True, True),      # 20-wide tiles, partial last channel block
    (torch.float32, 1, 12, 26, 8, 5, True, False, False),      # fp32
]


@pytest.mark.parametrize("case", DW_CASES, ids=lambda c: f"{str(c[0])[6:]}-C{c[4]}-k{c[5]}-{c[2]}x{c[3]}")
def test_depthwise_stencil(case, host_ops):
    dtype, B, H, W, C, k, use_bias, act, use_res = case
    x = _rnd(B, H, W, C, seed=5).to(dtype)
    w = _rnd(k * k, C, seed=6, scale=1.0 / k).to(dtype)
    bias = _rnd(C, seed=7, scale=0.2) if use_bias else None
    res = _rnd(B, H, W, C, seed=8).to(dtype) if use_res else None
    ref = emu_ops.dwconv2d(x, w, bias, k, act, residual=res)
    got = host_ops.dwconv2d(x, w, bias, k, act, residual=res)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert torch.allclose(got.float(), ref.float(), atol=tol, rtol=tol), float((got.float() - ref.float()).abs().max())
