"""Fused C3k2 block (csrc/c3k2f.hip: cv1 -> Bottleneck 3x3 / 3x3 + residual -> cv2 over [a | b | m], one kernel) on the CPU lane
emulator against the four-convolution composition it replaces, every stage rounded to bf16 as the unfused path stores it.  Map
sizes exercise the borders (tiles cut by the right / bottom edge, maps smaller than a tile, several tiles).  Shared with the GPU test."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

CASES = [(1, 8, 32), (1, 5, 7), (2, 19, 45), (1, 9, 70)]   # B, H, W


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def operands(case):
    from yolo_master_amd import ops

    B, H, W = case
    g = torch.Generator().manual_seed(H * 100 + W)
    bf = torch.bfloat16
    x = torch.randn(B, H, W, 64, generator=g).to(bf)
    ws = [torch.randn(64, 64, 1, 1, generator=g) * 64 ** -0.5, torch.randn(16, 32, 3, 3, generator=g) * 288 ** -0.5,
          torch.randn(32, 16, 3, 3, generator=g) * 144 ** -0.5, torch.randn(128, 96, 1, 1, generator=g) * 96 ** -0.5]
    bs = [torch.randn(w.shape[0], generator=g) * 0.3 for w in ws]
    packed = [ops.pack_conv_weight(w, bf) for w in ws]
    return x, ws, bs, packed


def reference(x, ws, bs):
    """torch fp32 arithmetic on bf16-rounded operands, each stage's output rounded to bf16."""
    bf = torch.bfloat16

    def conv(t, w, b, k, res=None):
        y = F.silu(F.conv2d(t.float().permute(0, 3, 1, 2), w.to(bf).float(), b, 1, k // 2)).permute(0, 2, 3, 1)
        if res is not None:
            y = res.float() + y
        return y.to(bf)

    y1 = conv(x, ws[0], bs[0], 1)
    b = y1[..., 32:]
    h = conv(b, ws[1], bs[1], 3)
    m = conv(h, ws[2], bs[2], 3, res=b)
    return conv(torch.cat([y1, m], -1), ws[3], bs[3], 1).float()


def run_case(lib, case, dev="cpu", stream=None):
    B, H, W = case
    x, ws, bs, packed = operands(case)
    ref = reference(x, ws, bs)
    bf = torch.bfloat16
    xb = torch.full((B, H, W, 64 + 16), 3.0, dtype=bf)
    xb[..., 8:72] = x
    xd = xb.to(dev)[..., 8:72]
    yb = torch.full((B, H, W, 128 + 8), 7.0, dtype=bf, device=dev)
    pk, bd = [w.to(dev) for w in packed], [b.to(dev) for b in bs]
    assert lib.ymk_c3k2_fused_supported(1, 64, 128, 32, 1, 0, 1) and not lib.ymk_c3k2_fused_supported(1, 128, 256, 64, 1, 0, 1)
    nchunk = lib.ymk_c3k2_fused_pool_chunks(H, W)
    assert nchunk == -(-H // 8) * -(-W // 32)
    part = torch.full((B, nchunk, 128), 9.0, dtype=torch.float32, device=dev)
    rc = lib.ymk_c3k2_fused_pooled(_p(xd), xd.stride(2), B, H, W, _p(pk[0]), pk[0].shape[1], _p(bd[0]), _p(pk[1]), pk[1].shape[1], _p(bd[1]),
                                   _p(pk[2]), pk[2].shape[1], _p(bd[2]), _p(pk[3]), pk[3].shape[1], _p(bd[3]), _p(yb), yb.stride(2), _p(part),
                                   None, stream)
    assert rc == 0
    got = yb[..., :128].float().cpu()
    # pooled partials: their sum over the chunks is the channel sum of the STORED output (what a global average pool of y reads)
    pooled = part.sum(1).cpu()
    want = got.double().sum((1, 2))
    assert float((pooled.double() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), "pooled partials != channel sums of y"
    y2 = torch.full_like(yb, 7.0)     # the entry point without pooling writes the same map
    assert lib.ymk_c3k2_fused(_p(xd), xd.stride(2), B, H, W, _p(pk[0]), pk[0].shape[1], _p(bd[0]), _p(pk[1]), pk[1].shape[1], _p(bd[1]),
                              _p(pk[2]), pk[2].shape[1], _p(bd[2]), _p(pk[3]), pk[3].shape[1], _p(bd[3]), _p(y2), y2.stride(2), stream) == 0
    assert torch.equal(y2.cpu(), yb.cpu())
    err = (got - ref).abs()
    scale = max(1.0, float(ref.abs().max()))
    # a stage's bf16 rounding may land on the other side of a boundary (fast SiLU, summation order): isolated one-ulp effects downstream
    assert float(err.max()) <= 4e-2 * scale, f"{case}: max err {float(err.max()):.3e}"
    assert float(err.mean()) <= 3e-3 * scale, f"{case}: mean err {float(err.mean()):.3e}"
    assert bool((yb[..., 128:].float().cpu() == 7.0).all()), "bytes between pixels were touched"
    return got


@pytest.mark.parametrize("case", CASES)
def test_c3k2_fused_on_emulator(case, hostlib):
    run_case(hostlib, case)
