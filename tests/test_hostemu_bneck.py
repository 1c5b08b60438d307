"""Fused 64-channel Bottleneck (csrc/bneck.hip: two 3x3 convolutions + shortcut as one kernel) on the CPU lane emulator against the two
convolutions it replaces (torch fp32 arithmetic on bf16 operands, the intermediate rounded to bf16).  Shared with the GPU test."""
import pytest
import torch
import torch.nn.functional as F

CASES = [(1, 8, 16, True), (2, 11, 21, True), (1, 5, 7, False), (1, 20, 20, True), (2, 9, 33, True)]   # B, H, W, shortcut


def run_case(ops, case, dev="cpu", sliced=False):
    B, H, W, add = case
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(H * 37 + W)
    xw = torch.randn(B, H, W, 128 if sliced else 64, generator=g).to(bf)
    x = xw[..., 64:] if sliced else xw                                        # a channel slice of a wider buffer (C3k's cat)
    w1, w2 = (torch.randn(64, 64, 3, 3, generator=g) * 576 ** -0.5 for _ in range(2))
    b1, b2 = (torch.randn(64, generator=g) * 0.2 for _ in range(2))
    t = x.float().permute(0, 3, 1, 2)
    h = F.silu(F.conv2d(t, w1.to(bf).float(), b1, padding=1)).to(bf).float()
    ref = F.silu(F.conv2d(h, w2.to(bf).float(), b2, padding=1))
    ref = (ref + t if add else ref).permute(0, 2, 3, 1)
    xd = xw.to(dev)
    xv = xd[..., 64:] if sliced else xd
    outw = torch.full((B, H, W, 96), 9.0, dtype=bf, device=dev)
    out = outw[..., 16:80]                                                      # ... and a slice as the destination
    p1, p2 = ops.pack_conv_weight(w1, bf).to(dev), ops.pack_conv_weight(w2, bf).to(dev)
    assert ops.bottleneck_fused_supported(bf, 64, 64, 64) and not ops.bottleneck_fused_supported(bf, 64, 32, 64)
    got = ops.bottleneck_fused(xv, p1, b1.to(dev), p2, b2.to(dev), add, out=out)
    assert got.data_ptr() == out.data_ptr()
    err = (got.float().cpu() - ref).abs()
    scale = max(1.0, float(ref.abs().max()))
    assert float(err.max()) <= 3e-2 * scale and float(err.mean()) <= 2e-3 * scale, f"{case}: max {float(err.max()):.3e} mean {float(err.mean()):.3e}"
    assert bool((outw[..., :16].float().cpu() == 9.0).all()) and bool((outw[..., 80:].float().cpu() == 9.0).all()), "wrote outside its channel slice"
    # against the two library convolutions it replaces: the same operands, roundings and accumulation order
    hh = ops.conv2d(xv, p1, b1.to(dev), 3, 1, True)
    two = ops.conv2d(hh, p2, b2.to(dev), 3, 1, True, residual=xv if add else None)
    d = (two.float().cpu() - got.float().cpu()).abs()
    assert float(d.mean()) <= 2e-4 * scale and float((d > 0).float().mean()) <= 0.02, f"{case}: differs from the unfused pair on {float((d > 0).float().mean()):.4f} of the elements"
    return got


@pytest.mark.parametrize("case", CASES)
def test_bottleneck_fused_on_emulator(case, hostlib, monkeypatch):
    from yolo_master_amd import ops

    monkeypatch.setattr(ops, "lib", hostlib)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    run_case(ops, case, sliced=case[1] == 11)
