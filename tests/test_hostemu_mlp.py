"""Fused ABlock MLP (csrc/mlp.hip: y = x + W2 SiLU(W1 x + b1) + b2, one kernel) on the CPU lane emulator against the two-convolution
composition it replaces, on the same bf16 operands (hidden tensor rounded to bf16 in both).  Shared with tests/test_gpu_kernels.py."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

CASES = [(64, 128, 100, 0, 0), (128, 256, 64, 0, 0), (128, 256, 203, 128, 64), (256, 512, 130, 0, 256)]   # C, hidden, tokens, x pad, y pad


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def run_case(lib, case, dev="cpu", stream=None):
    Cc, Hd, M, xpad, ypad = case
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(Cc + M)
    x = torch.randn(M, Cc, generator=g).to(bf)
    w1 = (torch.randn(Hd, Cc, generator=g) * Cc ** -0.5).to(bf)
    w2 = (torch.randn(Cc, Hd, generator=g) * Hd ** -0.5).to(bf)
    b1, b2 = torch.randn(Hd, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.2
    h = F.silu(x.float() @ w1.float().t() + b1).to(bf)                      # what the unfused pair stores between the two convs
    ref = x.float() + h.float() @ w2.float().t() + b2
    xb = torch.full((M, Cc + xpad), 3.0, dtype=bf)
    xb[:, xpad // 2: xpad // 2 + Cc] = x
    xd = xb.to(dev)[:, xpad // 2: xpad // 2 + Cc]
    yb = torch.full((M, Cc + ypad), 7.0, dtype=bf, device=dev)
    y = yb[:, ypad // 2: ypad // 2 + Cc]
    w1d, w2d, b1d, b2d = w1.to(dev), w2.to(dev), b1.to(dev), b2.to(dev)
    assert lib.ymk_mlp_fused_supported(1, Cc, Hd)
    rc = lib.ymk_mlp_fused(_p(xd), xd.stride(0), _p(w1d), Cc, _p(b1d), _p(w2d), Hd, _p(b2d), _p(y), y.stride(0), M, Cc, Hd, stream)
    assert rc == 0
    got = y.float().cpu()
    err = (got - ref).abs()
    assert float(err.max()) <= 3e-2 * max(1.0, float(ref.abs().max())), f"{case}: max err {float(err.max()):.3e}"
    assert float(err.mean()) <= 4e-3, f"{case}: mean err {float(err.mean()):.3e}"    # bf16 rounding of the output only
    if ypad:
        assert bool((yb[:, : ypad // 2].float().cpu() == 7.0).all()) and bool((yb[:, ypad // 2 + Cc:].float().cpu() == 7.0).all())


@pytest.mark.parametrize("case", CASES)
def test_mlp_fused_on_emulator(case, hostlib):
    run_case(hostlib, case)


PROJ_CASES = [(128, 256, 64, 0, 0), (128, 256, 203, 128, 64), (256, 512, 130, 0, 256)]   # C, hidden, tokens, a / x pad, y pad


def run_proj_case(lib, case, dev="cpu", stream=None):
    """ymk_proj_mlp_fused (x1 = x + Wp a + bp; y = x1 + W2 SiLU(W1 x1 + b1) + b2) against the composition it replaces on the same 16-bit
    operands: x1 and the hidden tensor rounded to bf16 where the unfused kernels store them."""
    Cc, Hd, M, xpad, ypad = case
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(3 * Cc + M)
    a = torch.randn(M, Cc, generator=g).to(bf)
    x = torch.randn(M, Cc, generator=g).to(bf)
    wp = (torch.randn(Cc, Cc, generator=g) * Cc ** -0.5).to(bf)
    w1 = (torch.randn(Hd, Cc, generator=g) * Cc ** -0.5).to(bf)
    w2 = (torch.randn(Cc, Hd, generator=g) * Hd ** -0.5).to(bf)
    bp, b1, b2 = torch.randn(Cc, generator=g) * 0.2, torch.randn(Hd, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.2
    x1 = (x.float() + (a.float() @ wp.float().t() + bp)).to(bf)             # what the unfused projection stores
    h = F.silu(x1.float() @ w1.float().t() + b1).to(bf)
    ref = x1.float() + h.float() @ w2.float().t() + b2

    def padded(t, pad, fill):
        buf = torch.full((M, Cc + pad), fill, dtype=bf)
        buf[:, pad // 2: pad // 2 + Cc] = t
        return buf.to(dev)[:, pad // 2: pad // 2 + Cc]

    ad, xd = padded(a, xpad, 3.0), padded(x, xpad, 5.0)
    yb = torch.full((M, Cc + ypad), 7.0, dtype=bf, device=dev)
    y = yb[:, ypad // 2: ypad // 2 + Cc]
    d = [t.to(dev) for t in (wp, bp, w1, b1, w2, b2)]
    rc = lib.ymk_proj_mlp_fused(_p(ad), ad.stride(0), _p(d[0]), Cc, _p(d[1]), _p(xd), xd.stride(0), _p(d[2]), Cc, _p(d[3]), _p(d[4]), Hd,
                                _p(d[5]), _p(y), y.stride(0), M, Cc, Hd, stream)
    assert rc == 0
    got = y.float().cpu()
    err = (got - ref).abs()
    # x1's rounding may land on the other side of a bf16 boundary in a few elements (fp32 summation order of the projection): one bf16 ulp of
    # x1 (2^-8 relative) propagates through the MLP
    assert float(err.max()) <= 4e-2 * max(1.0, float(ref.abs().max())), f"{case}: max err {float(err.max()):.3e}"
    assert float(err.mean()) <= 5e-3, f"{case}: mean err {float(err.mean()):.3e}"
    if ypad:
        assert bool((yb[:, : ypad // 2].float().cpu() == 7.0).all()) and bool((yb[:, ypad // 2 + Cc:].float().cpu() == 7.0).all())
    assert lib.ymk_proj_mlp_fused(_p(ad), ad.stride(0), _p(d[0]), Cc, _p(d[1]), _p(xd), xd.stride(0), _p(d[2]), Cc, _p(d[3]), _p(d[4]), Hd,
                                  _p(d[5]), _p(y), y.stride(0), M, 64, 128, stream) != 0, "C = 64 has no projection variant"


@pytest.mark.parametrize("case", PROJ_CASES)
def test_proj_mlp_fused_on_emulator(case, hostlib):
    run_proj_case(hostlib, case)
