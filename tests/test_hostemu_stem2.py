"""Fused stem + row-1 convolution (csrc/stem2.hip) on the CPU lane emulator against the two-layer composition it replaces:
fp32 stem (3 -> 32, 3x3/s2, SiLU) rounded to bf16, then 32 -> 64 3x3/s2 + SiLU on bf16 operands.  Image sizes exercise the
borders: odd stem / output sizes, tiles cut by the right and bottom edges, several tiles per row.  Shared with the GPU test."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

CASES = [(1, 16, 16), (2, 36, 140), (1, 70, 44), (1, 23, 264)]   # B, H, W (W % 4 == 0)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def run_case(lib, case, dev="cpu", stream=None):
    from yolo_master_amd import ops

    B, H, W = case
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.rand(B, 3, H, W, generator=g)
    w0 = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    b0 = torch.randn(32, generator=g) * 0.3
    w1 = torch.randn(64, 32, 3, 3, generator=g) * (9 * 32) ** -0.5
    b1 = torch.randn(64, generator=g) * 0.2
    bf = torch.bfloat16
    h = F.silu(F.conv2d(x, w0, b0, 2, 1)).to(bf)                                   # the stem map as the unfused path stores it
    ref = F.silu(F.conv2d(h.float(), w1.to(bf).float(), b1, 2, 1)).permute(0, 2, 3, 1)
    wt0 = w0.permute(0, 2, 3, 1).reshape(32, 27).t().contiguous()                  # [27][32], k = (ky, kx, c)
    w1p = ops.pack_conv_weight(w1, bf)
    H2, W2 = ref.shape[1:3]
    ldy = 64 + 8
    yb = torch.full((B, H2, W2, ldy), 7.0, dtype=bf, device=dev)
    xd, wt0d, b0d, w1d, b1d = x.to(dev), wt0.to(dev), b0.to(dev), w1p.to(dev), b1.to(dev)
    assert lib.ymk_stem_pair_supported(1, 3, 32, 64, 3, 2, 3, 2) and not lib.ymk_stem_pair_supported(1, 3, 16, 32, 3, 2, 3, 2)
    rc = lib.ymk_stem_pair(_p(xd), B, H, W, _p(wt0d), _p(b0d), 32, _p(w1d), w1p.shape[1], _p(b1d), 64, _p(yb), ldy, stream)
    assert rc == 0
    got = yb[..., :64].float().cpu()
    err = (got - ref).abs()
    # the stem map may differ by one bf16 ulp where the fast SiLU / MFMA summation order straddles a rounding boundary
    assert float(err.max()) <= 3e-2 * max(1.0, float(ref.abs().max())), f"{case}: max err {float(err.max()):.3e}"
    assert float(err.mean()) <= 3e-3, f"{case}: mean err {float(err.mean()):.3e}"
    assert bool((yb[..., 64:].float().cpu() == 7.0).all()), "bytes between pixel rows were touched"
    return got


@pytest.mark.parametrize("case", CASES)
def test_stem_pair_on_emulator(case, hostlib):
    """Default variant: fp32 stem operands as two bf16 parts each (hi*hi + hi*lo + lo*hi on the bf16 matrix cores)."""
    run_case(hostlib, case)


def test_bf16_split_carries_fp32_products():
    """The split the kernel uses (v = hi + lo, both bf16, lo*lo dropped): relative error of a 27-term dot product of image values in
    [0, 1] with weights ~N(0, 0.3) stays below 2^-15 of the sum of magnitudes — two orders under the bf16 rounding of the stem map."""
    g = torch.Generator().manual_seed(0)
    x = torch.rand(4096, 27, generator=g)
    w = torch.randn(4096, 27, generator=g) * 0.3
    bf = torch.bfloat16

    def split(v):
        hi = v.to(bf).float()
        return hi, (v - hi).to(bf).float()

    xh, xl = split(x)
    wh, wl = split(w)
    got = (xh * wh + xh * wl + xl * wh).double().sum(1)
    ref = (x.double() * w.double()).sum(1)
    scale = (x.double() * w.double()).abs().sum(1)
    assert float(((got - ref).abs() / scale).max()) < 2.0 ** -15
