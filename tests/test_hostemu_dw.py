"""Matrix-core depthwise convolution (csrc/dwmfma.hip: Toeplitz MFMA formulation) on the CPU lane emulator (tests/hostemu):
`ymk_dw_toeplitz_pack` + `ymk_dwconv2d_mfma` / `ymk_esmoe_dw_mfma` through their ctypes bindings against torch's depthwise
convolution on the same bf16 inputs — every filter size, ragged tiles, channel-slice views, bias / SiLU / residual, and the
CSR-driven expert dispatch.  The same cases run on the GPU in tests/test_gpu_kernels.py (run_dw_case / run_moe_dw_case)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _view(t, pad, dev):
    """The values of t [B,H,W,C] as a channel slice of a wider buffer on `dev`."""
    buf = torch.full((*t.shape[:3], t.shape[3] + pad), 3.0, dtype=t.dtype)
    buf[..., pad // 2: pad // 2 + t.shape[3]] = t
    return buf.to(dev)[..., pad // 2: pad // 2 + t.shape[3]]


def toeplitz(lib, w_kkc, k, dev, stream=None):
    Cc = w_kkc.shape[1]
    assert lib.ymk_dw_mfma_supported(1, Cc, k)
    out = torch.empty((lib.ymk_dw_toeplitz_elems(Cc, k),), dtype=torch.bfloat16, device=dev)
    w_d = w_kkc.to(dev)
    assert lib.ymk_dw_toeplitz_pack(_p(w_d), Cc, k, _p(out), stream) == 0
    if dev != "cpu":
        torch.cuda.synchronize()   # w_d is released on return
    return out


DW_CASES = [
    # B, H, W, C, k, bias, act, residual, xpad, ypad
    (2, 16, 32, 16, 3, True, True, False, 0, 0),       # exactly one tile
    (1, 21, 37, 32, 5, False, False, True, 16, 0),     # ragged in both directions, two channel blocks, residual, input view
    (2, 20, 20, 16, 7, True, False, True, 0, 16),      # AAttn.pe shape class (7x7 + residual), output view
    (1, 33, 70, 16, 9, False, True, False, 0, 0),      # three x tiles, three y tiles, largest filter
    (3, 8, 8, 48, 3, True, True, False, 0, 0),         # map smaller than a tile, three channel blocks
]


def run_dw_case(lib, case, dev="cpu", stream=None):
    B, H, W, Cc, k, use_bias, act, use_res, xpad, ypad = case
    bf = torch.bfloat16
    x = _rnd(B, H, W, Cc, seed=1).to(bf)
    w = _rnd(Cc, 1, k, k, seed=2, scale=1.0 / k).to(bf)
    bias = _rnd(Cc, seed=3, scale=0.2) if use_bias else None
    res = _rnd(B, H, W, Cc, seed=4).to(bf) if use_res else None
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, 1, k // 2, 1, Cc)
    if act:
        ref = F.silu(ref)
    if use_res:
        ref = ref + res.float().permute(0, 3, 1, 2)
    ref = ref.permute(0, 2, 3, 1)
    w_kkc = w.reshape(Cc, k * k).t().contiguous()
    tp = toeplitz(lib, w_kkc, k, dev, stream)
    xd = _view(x, xpad, dev)
    rd = None if res is None else _view(res, 32, dev)
    ybuf = torch.full((B, H, W, Cc + ypad), 7.0, dtype=bf, device=dev)
    y = ybuf[..., ypad // 2: ypad // 2 + Cc]
    bias_d = None if bias is None else bias.to(dev)
    rc = lib.ymk_dwconv2d_mfma(_p(xd), _p(tp), _p(bias_d), _p(rd), _p(y), B, H, W, Cc, k, xd.stride(2),
                               y.stride(2), rd.stride(2) if use_res else 0, 1 if act else 0, stream)
    assert rc == 0
    got = y.float().cpu()
    err = (got - ref).abs()
    assert float(err.max()) <= 2e-2 * max(1.0, float(ref.abs().max())), f"{case}: max err {float(err.max()):.3e}"
    assert float(err.mean()) <= 2e-3, f"{case}: mean err {float(err.mean()):.3e}"     # bf16 output rounding only
    if ypad:   # nothing outside the channel slice was touched
        assert bool((ybuf[..., : ypad // 2].float().cpu() == 7.0).all()) and bool((ybuf[..., ypad // 2 + Cc:].float().cpu() == 7.0).all())


MOE_CASES = [
    # B, H, W, C, ksizes, sel rows (expert per slot, -1 = dropped)
    (4, 18, 35, 16, [3, 5, 7, 9], [[0, 3], [1, -1], [2, 3], [0, 1]]),
    (3, 16, 32, 32, [3, 5, 7, 9], [[3, -1], [3, -1], [2, -1]]),          # one expert unused, single slot per image
    (5, 9, 12, 16, [3, 5, 7], [[0, 1], [0, 2], [1, 2], [0, -1], [2, -1]]),
]


def run_moe_dw_case(lib, case, dev="cpu", stream=None):
    B, H, W, Cc, ks, sel = case
    bf = torch.bfloat16
    E, top_k = len(ks), len(sel[0])
    x = _rnd(B, H, W, Cc, seed=5).to(bf)
    ws = [_rnd(Cc, 1, k, k, seed=10 + e, scale=1.0 / k).to(bf) for e, k in enumerate(ks)]
    tp = torch.cat([toeplitz(lib, w.reshape(Cc, k * k).t().contiguous(), k, dev, stream) for w, k in zip(ws, ks)])
    pairs = [[] for _ in range(E)]
    for b in range(B):
        for s, e in enumerate(sel[b]):
            if e >= 0:
                pairs[e].append(b * top_k + s)
    off = [0]
    for e in range(E):
        off.append(off[-1] + len(pairs[e]))
    flat = [p for e in range(E) for p in pairs[e]] + [0] * (B * top_k - off[-1])
    i32 = dict(dtype=torch.int32, device=dev)
    out = torch.full((B * top_k, H, W, Cc), 5.0, dtype=bf, device=dev)
    xd = x.to(dev)
    ks_d, off_d, flat_d = torch.tensor(ks, **i32), torch.tensor(off, **i32), torch.tensor(flat, **i32)   # kept alive across the call
    rc = lib.ymk_esmoe_dw_mfma(_p(xd), B, H, W, Cc, Cc, _p(tp), _p(ks_d), sum({1 << (k // 2) for k in ks}), E, top_k, _p(off_d), _p(flat_d), _p(out), stream)
    assert rc == 0
    got = out.float().cpu()
    for b in range(B):
        for s, e in enumerate(sel[b]):
            if e < 0:
                assert bool((got[b * top_k + s] == 5.0).all()), "a dropped slot's plane was written"
                continue
            ref = F.conv2d(x[b: b + 1].float().permute(0, 3, 1, 2), ws[e].float(), None, 1, ks[e] // 2, 1, Cc)[0].permute(1, 2, 0)
            err = (got[b * top_k + s] - ref).abs()
            assert float(err.max()) <= 2e-2 * max(1.0, float(ref.abs().max())), f"{case} image {b} slot {s}: {float(err.max()):.3e}"


@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv_mfma_on_emulator(case, hostlib):
    run_dw_case(hostlib, case)


@pytest.mark.parametrize("case", MOE_CASES)
def test_esmoe_dw_mfma_on_emulator(case, hostlib):
    run_moe_dw_case(hostlib, case)
