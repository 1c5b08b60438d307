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
