"""Config-5 rows on the GPU: the first-implementation kernels of include/ymk_mixture.h through the C-ABI, against
(a) the torch restatement of each entry point's contract (tests/emu_ops.py) on the same seeded inputs and
(b) the REAL reference's golden vectors for the modules and the whole config-5 detector.

First hardware run: round 2 (profiles/r02_first_hw_run.log); part of the driver's `pytest -m gpu` run since then.
"""
import os

import numpy as np
import pytest
import torch

from tests import emu_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.6e-2}


def _rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def _view(t, pad):
    """The same values as a channel slice of a wider buffer (pixel stride > C)."""
    if not pad:
        return t.to(DEV)
    wide = torch.zeros((*t.shape[:3], t.shape[3] + pad), dtype=t.dtype, device=DEV)
    wide[..., : t.shape[3]] = t.to(DEV)
    return wide[..., : t.shape[3]]


def _cmp(got, ref, dtype, what):
    got, ref = got.float().cpu(), ref.float()
    err = float((got - ref).abs().max())
    assert err <= TOL[dtype] * max(1.0, float(ref.abs().max())), f"{what}: max |d| {err:.3e} (|ref| max {float(ref.abs().max()):.3f})"


DTYPES = [torch.float32, torch.bfloat16]


@pytest.mark.parametrize("dtype", DTYPES)
def test_norms_and_elementwise(dtype):
    from yolo_master_amd import ops

    x = _rnd(3, 9, 11, 48, seed=1, dtype=dtype)
    w, b = 1.0 + 0.1 * _rnd(48, seed=2), 0.1 * _rnd(48, seed=3)
    for groups, act, pad in ((8, False, 0), (3, "silu", 16), (1, "silu", 0)):
        _cmp(ops.group_norm(_view(x, pad), groups, w.to(DEV), b.to(DEV), 1e-5, act=act), emu_ops.group_norm(x, groups, w, b, 1e-5, act=act),
             dtype, f"group_norm g{groups}")
    res = _rnd(3, 9, 11, 48, seed=4, dtype=dtype)
    _cmp(ops.group_norm(x.to(DEV), 8, w.to(DEV), b.to(DEV), 1e-5, residual=res.to(DEV)),
         emu_ops.group_norm(x, 8, w, b, 1e-5, residual=res), dtype, "group_norm + residual")
    rows_w, rows_b = 1.0 + 0.1 * _rnd(5, 48, seed=5), 0.1 * _rnd(5, 48, seed=6)
    rows = torch.tensor([4, 0, 2], dtype=torch.int32)
    _cmp(ops.group_norm(x.to(DEV), 8, rows_w.to(DEV), rows_b.to(DEV), 1e-5, act="silu", affine_rows=rows.to(DEV)),
         emu_ops.group_norm(x, 8, rows_w, rows_b, 1e-5, act="silu", affine_rows=rows), dtype, "group_norm affine rows")
    _cmp(ops.group_norm(x.to(DEV), 8, None, None, 1e-5), emu_ops.group_norm(x, 8, None, None, 1e-5), dtype, "group_norm no affine")
    # a slab large enough to be split over several workgroups (chunk partials combined exactly, in chunk order): vector and scalar kernels
    xb = _rnd(1, 96, 96, 16, seed=30, dtype=dtype) * 0.5 + 1.0
    _cmp(ops.group_norm(xb.to(DEV), 2, w[:16].to(DEV), b[:16].to(DEV), 1e-5), emu_ops.group_norm(xb, 2, w[:16], b[:16], 1e-5), dtype, "group_norm chunked")
    xb6 = _rnd(1, 128, 90, 6, seed=31) * 0.5 - 1.0
    _cmp(ops.group_norm(xb6.to(DEV), 1, w[:6].to(DEV), b[:6].to(DEV), 1e-5, act="silu"), emu_ops.group_norm(xb6, 1, w[:6], b[:6], 1e-5, act="silu"),
         torch.float32, "group_norm chunked C=6")
    if dtype == torch.float32:   # mean^2 >> variance (a sum-of-squares formula loses the variance; torch's own fp32 group_norm is 4e-3 off here): vs fp64
        xo = _rnd(1, 96, 96, 16, seed=32) * 0.05 + 6.0
        xd = xo.double().reshape(1, 96 * 96, 2, 8)
        r64 = ((xd - xd.mean((1, 3), keepdim=True)) / (xd.var((1, 3), unbiased=False, keepdim=True) + 1e-5).sqrt()).reshape(1, 96, 96, 16)
        got = ops.group_norm(xo.to(DEV), 2, None, None, 1e-5).cpu().double()
        assert float((got - r64).abs().max()) <= 5e-5, float((got - r64).abs().max())
    x6 = _rnd(2, 5, 7, 6, seed=7)                      # odd channel count inside an 8-wide buffer, fp32 -> fp32 (router path)
    buf = torch.zeros((2, 5, 7, 8), device=DEV)
    ops.group_norm(_view(x6, 2), 3, w[:6].to(DEV), b[:6].to(DEV), 1e-5, act="silu", out=buf[..., :6])
    _cmp(buf[..., :6], emu_ops.group_norm(x6, 3, w[:6], b[:6], 1e-5, act="silu"), torch.float32, "group_norm C=6")
    assert float(buf[..., 6:].abs().max()) == 0.0
    _cmp(ops.layer_norm(_view(x, 8), w.to(DEV), b.to(DEV), 1e-5), emu_ops.layer_norm(x, w, b, 1e-5), dtype, "layer_norm")
    # register-resident rows: 256 channels = 32 lanes per token in 16-bit (two tokens per wave), 64 in fp32; 35 tokens (ragged last group);
    # 520 channels (65 / 130 vectors: the looping kernel)
    for cw, sd in ((256, 50), (520, 53)):
        xw, ww, bw = _rnd(1, 5, 7, cw, seed=sd, dtype=dtype), 1.0 + 0.1 * _rnd(cw, seed=sd + 1), 0.1 * _rnd(cw, seed=sd + 2)
        _cmp(ops.layer_norm(xw.to(DEV), ww.to(DEV), bw.to(DEV), 1e-5), emu_ops.layer_norm(xw, ww, bw, 1e-5), dtype, f"layer_norm C={cw}")
    y = _rnd(3, 9, 11, 48, seed=8, dtype=dtype)
    _cmp(ops.eltwise_mul(x.to(DEV), _view(y, 8)), emu_ops.eltwise_mul(x, y), dtype, "mul")
    _cmp(ops.eltwise_mul(x.to(DEV), y.to(DEV), act_a="sigmoid"), emu_ops.eltwise_mul(x, y, act_a="sigmoid"), dtype, "sigmoid-mul")
    _cmp(ops.lerp(x.to(DEV), y.to(DEV), 0.3), emu_ops.lerp(x, y, 0.3), dtype, "lerp")
    gate = torch.sigmoid(_rnd(3, 1, 1, 48, seed=9))
    _cmp(ops.channel_gate(x.to(DEV), gate.to(DEV)), emu_ops.channel_gate(x, gate), dtype, "channel_gate")
    _cmp(ops.fma_gate(x.to(DEV), y.to(DEV), gate.to(DEV), 0.37), emu_ops.fma_gate(x, y, gate, 0.37), dtype, "fma_gate per image")
    _cmp(ops.fma_gate(x.to(DEV), x.to(DEV), y.to(DEV), -0.2), emu_ops.fma_gate(x, x, y, -0.2), dtype, "fma_gate map")
    wts = torch.softmax(_rnd(3, 9, 11, 3, seed=10), -1)
    parts = [_rnd(3, 9, 11, 48, seed=11 + i, dtype=dtype) for i in range(3)]
    _cmp(ops.weighted_sum(wts.to(DEV), [p.to(DEV) for p in parts]), emu_ops.weighted_sum(wts, parts), dtype, "weighted_sum tokens")
    wimg = torch.softmax(_rnd(3, 1, 1, 2, seed=14), -1)
    _cmp(ops.weighted_sum(wimg.to(DEV), [p.to(DEV) for p in parts[:2]]), emu_ops.weighted_sum(wimg, parts[:2]), dtype, "weighted_sum images")
    # channel counts below the 16-byte vector width take the scalar kernels
    xs, ys = _rnd(2, 5, 7, 6, seed=40, dtype=dtype), _rnd(2, 5, 7, 6, seed=41, dtype=dtype)
    gs = torch.sigmoid(_rnd(2, 1, 1, 6, seed=42))
    _cmp(ops.layer_norm(xs.to(DEV), w[:6].to(DEV), b[:6].to(DEV), 1e-5), emu_ops.layer_norm(xs, w[:6], b[:6], 1e-5), dtype, "layer_norm C=6")
    _cmp(ops.eltwise_mul(xs.to(DEV), ys.to(DEV), act_a="sigmoid"), emu_ops.eltwise_mul(xs, ys, act_a="sigmoid"), dtype, "sigmoid-mul C=6")
    _cmp(ops.lerp(xs.to(DEV), ys.to(DEV), 0.7), emu_ops.lerp(xs, ys, 0.7), dtype, "lerp C=6")
    _cmp(ops.channel_gate(xs.to(DEV), gs.to(DEV)), emu_ops.channel_gate(xs, gs), dtype, "channel_gate C=6")
    _cmp(ops.fma_gate(xs.to(DEV), ys.to(DEV), gs.to(DEV), 0.5), emu_ops.fma_gate(xs, ys, gs, 0.5), dtype, "fma_gate C=6")
    _cmp(ops.fma_gate(xs.to(DEV), xs.to(DEV), ys.to(DEV), 0.5), emu_ops.fma_gate(xs, xs, ys, 0.5), dtype, "fma_gate map C=6")
    ws2 = torch.softmax(_rnd(2, 5, 7, 2, seed=43), -1)
    _cmp(ops.weighted_sum(ws2.to(DEV), [xs.to(DEV), ys.to(DEV)]), emu_ops.weighted_sum(ws2, [xs, ys]), dtype, "weighted_sum C=6")
    mixed = torch.sigmoid(_rnd(3, 9, 11, 48, seed=44))                      # fp32 gate map over activations of the compute dtype
    _cmp(ops.fma_gate(x.to(DEV), y.to(DEV), mixed.to(DEV), 0.25), emu_ops.fma_gate(x, y, mixed, 0.25), dtype, "fma_gate fp32 map")
    _cmp(ops.group_norm(x.to(DEV), 8, w.to(DEV), b.to(DEV), 1e-5, out_dtype=torch.float32), emu_ops.group_norm(x, 8, w, b, 1e-5, out_dtype=torch.float32),
         torch.float32 if dtype == torch.float32 else dtype, "group_norm -> fp32")
    for act in ("sigmoid", "gelu"):
        wp = (0.2 * _rnd(16, 64, seed=15)).to(dtype)
        wp[:, 48:] = 0
        bias = 0.1 * _rnd(16, seed=16)
        _cmp(ops.conv2d_act(x.to(DEV), wp.to(DEV), bias.to(DEV), 1, 1, act), emu_ops.conv2d_act(x, wp, bias, 1, 1, act), dtype, f"conv {act}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_pools_stats_shuffle_gather(dtype):
    from yolo_master_amd import ops

    x = _rnd(2, 13, 10, 32, seed=20, dtype=dtype)
    for ho, wo in ((6, 5), (3, 2), (1, 1), (13, 10)):
        _cmp(ops.adaptive_avg_pool(_view(x, 8), ho, wo), emu_ops.adaptive_avg_pool(x, ho, wo), dtype, f"adaptive pool {ho}x{wo}")
    _cmp(ops.avg_pool(x.to(DEV), 4, out_dtype=torch.float32), emu_ops.avg_pool(x, 4, out_dtype=torch.float32), torch.float32, "avg_pool 4")
    _cmp(ops.avg_pool(x.to(DEV), 1, out_dtype=torch.float32), x.float(), torch.float32, "avg_pool 1 (cast)")
    _cmp(ops.channel_stats(_view(x, 8)), emu_ops.channel_stats(x), torch.float32, "mean")
    _cmp(ops.channel_stats(x.to(DEV), want_std=True), emu_ops.channel_stats(x, want_std=True), torch.float32, "mean | std")
    x100 = _rnd(2, 3, 3, 100, seed=21, dtype=dtype)                    # more than one 64-channel block, ragged
    _cmp(ops.channel_stats(x100.to(DEV), want_std=True), emu_ops.channel_stats(x100, want_std=True), torch.float32, "stats C=100")
    xbig = _rnd(2, 60, 70, 16, seed=25, dtype=dtype) + 0.5      # 4200 pixels: five pixel chunks per image, partials combined in chunk order
    _cmp(ops.channel_stats(xbig.to(DEV), want_std=True), emu_ops.channel_stats(xbig, want_std=True), torch.float32, "stats chunked")
    _cmp(ops.channel_stats(_view(xbig, 8)), emu_ops.channel_stats(xbig), torch.float32, "mean chunked")
    parts = [x, _rnd(2, 6, 5, 32, seed=22, dtype=dtype), _rnd(2, 3, 2, 32, seed=23, dtype=dtype)]
    _cmp(ops.mean_upsampled([p.to(DEV) for p in parts]), emu_ops.mean_upsampled(parts), dtype, "mean_upsampled")
    a, b = _rnd(2, 7, 5, 24, seed=24, dtype=dtype), _rnd(2, 7, 5, 40, seed=25, dtype=dtype)
    for groups in (1, 2, 4):
        got = ops.channel_shuffle_cat([_view(a, 8), b.to(DEV)], groups)
        assert torch.equal(got.cpu(), emu_ops.channel_shuffle_cat([a, b], groups)), f"shuffle groups {groups}"
    a2 = _rnd(2, 7, 5, 40, seed=28, dtype=dtype)                     # equal halves, two groups: the vectorised even / odd interleave
    assert torch.equal(ops.channel_shuffle_cat([a2.to(DEV), _view(b, 8)], 2).cpu(), emu_ops.channel_shuffle_cat([a2, b], 2))
    idx = torch.tensor([[2, 0], [1, 3]], dtype=torch.int32)
    for k, cin in ((3, 16), (1, 32)):
        xe = _rnd(2, 6, 7, cin, seed=26, dtype=dtype)
        wp = torch.zeros((4, 8, (k * k * cin + 63) // 64 * 64), dtype=dtype)
        wp[:, :, : k * k * cin] = (_rnd(4, 8, k * k * cin, seed=27) * (k * k * cin) ** -0.5).to(dtype)
        _cmp(ops.expert_conv(xe.to(DEV), wp.to(DEV), k, idx.to(DEV)), emu_ops.expert_conv(xe, wp, k, idx), dtype, f"expert_conv k{k}")


def test_router_tails():
    from yolo_master_amd import ops

    logits = _rnd(3, 10, 12, 4, seed=30, scale=2.0)
    for n, k, it in ((3, 0, 1.0), (3, 2, 1.25), (3, 1, 0.5), (4, 3, 1.0)):
        w, active = ops.token_softmax(logits.to(DEV), n, it, top_k=k)
        rw, ra = emu_ops.token_softmax(logits, n, it, top_k=k)
        assert torch.equal(w.cpu() > 0, rw > 0), f"selected experts n={n} k={k}"
        _cmp(w, rw, torch.float32, f"token_softmax n={n} k={k}")
        assert torch.equal(active.cpu(), ra)
    lg = _rnd(2, 5, 5, 4, seed=31)
    lg[0, :, :, 2] = -60.0                                             # expert 2 never selected in image 0
    _, active = ops.token_softmax(lg.to(DEV), 3, 1.0, top_k=2)
    assert int(active[0, 2]) == 0 and int(active[1].sum()) >= 2
    for B, E, k, bias in ((5, 4, 2, 20.0), (3, 8, 2, -20.0), (4, 16, 2, 0.0), (3, 6, 3, 0.3), (2, 4, 1, 0.0)):
        g, loc = _rnd(B, 1, 1, E, seed=32 + B, scale=2.0), _rnd(B, 1, 1, E, seed=33 + E)
        cp = _rnd(B, 1, 1, 1, seed=34) + bias
        w, idx, probs, rows = ops.gated_route_decide(g.to(DEV), loc.to(DEV), 0.4, 1 / 1.2, k, cp.to(DEV))
        rw, ridx, rprobs, rrows = emu_ops.gated_route_decide(g, loc, 0.4, 1 / 1.2, k, cp)
        assert torch.equal(idx.cpu(), ridx) and torch.equal(rows.cpu(), rrows), f"routed experts B={B} E={E}"
        _cmp(w, rw, torch.float32, "gate weights")
        _cmp(probs, rprobs, torch.float32, "gate probs")
    # logits beyond +-30 (round-3 advisor finding): the three clamp conventions differ there — clamp then / T (gated.py:141-142),
    # / T then clamp (gated.py:972), no clamp at all (a router's own nn.Softmax gated.py:958, routers.py:207 `_process_logits`)
    g, loc = torch.zeros(4, 1, 1, 6), _rnd(4, 1, 1, 6, seed=35, scale=4.0).clamp(-12.0, 12.0)
    loc[:, 0, 0, 1], loc[:, 0, 0, 4] = 45.0, -50.0     # one logit per row beyond each bound (several would tie at the bound: top-k order undefined)
    cp = torch.full((4, 1, 1, 1), 100.0)
    seen = []
    for mode in (0, 1, 2):
        w, idx, probs, rows = ops.gated_route_decide(g.to(DEV), loc.to(DEV), -100.0, 2.0, 2, cp.to(DEV), clamp=mode)
        rw, ridx, rprobs, rrows = emu_ops.gated_route_decide(g, loc, -100.0, 2.0, 2, cp, clamp=mode)
        assert torch.equal(idx.cpu(), ridx), f"clamp mode {mode}"
        _cmp(w, rw, torch.float32, f"gate weights, clamp mode {mode}")
        _cmp(probs, rprobs, torch.float32, f"gate probs, clamp mode {mode}")
        seen.append(rprobs)
    assert float(seen[1][:, 0].min() / seen[0][:, 0].max()) > 1e3 and float(seen[2][:, 0].min() / seen[1][:, 0].max()) > 1e3, \
        "the test logits must reach the clamp"


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_family(dtype):
    from yolo_master_amd import ops

    for heads, hd, (H, W), (Hk, Wk) in ((2, 16, (9, 11), (9, 11)), (8, 8, (14, 18), (14, 18)), (1, 16, (12, 20), (6, 10)), (2, 32, (5, 7), (2, 3)),
                                        (1, 64, (8, 8), (8, 8)), (3, 24, (6, 5), (6, 5)),
                                        # matrix-core kernel (16-bit, head_dim 16 / 32 / 64): several 64-key blocks (online softmax), ragged last block,
                                        # more than 256 queries (two workgroups per (image, head))
                                        (2, 32, (20, 24), (10, 13)), (1, 16, (30, 30), (16, 16)), (1, 64, (18, 18), (9, 9)), (3, 32, (17, 16), (17, 16))):
        c = heads * hd
        qkv = _rnd(2, H, W, 3 * c, seed=40 + hd, dtype=dtype)
        kv = _rnd(2, Hk, Wk, 2 * c, seed=41 + hd, dtype=dtype)
        qd, kd = qkv.to(DEV), kv.to(DEV)
        _cmp(ops.attention(qd[..., :c], kd[..., :c], kd[..., c:], heads, hd, hd ** -0.5),
             emu_ops.attention(qkv[..., :c], kv[..., :c], kv[..., c:], heads, hd, hd ** -0.5), dtype, f"attention h{heads} d{hd} {H}x{W}/{Hk}x{Wk}")
    if dtype != torch.float32:
        # what the MoT full-attention expert relies on for 16-bit maps of more than 1024 tokens (nn/mixture.py): whole-map attention through the
        # [Q | K | V] area-attention entry point and through ymk_attention on separate q / k / v views are the same kernel — identical results
        heads, hd, H, W = 2, 32, 36, 30
        c = heads * hd
        qkv = _rnd(1, H, W, 3 * c, seed=47, dtype=dtype).to(DEV)
        v_sep = qkv[..., 2 * c:].clone()
        a = ops.area_attn(qkv, heads, 1)
        b = ops.attention(qkv[..., :c], qkv[..., c:2 * c], v_sep, heads, hd, hd ** -0.5)
        if DEV == "cpu":   # lane-emulator run: area_attn is the torch restatement of the v0 contract there, not the kernel
            _cmp(b, a, dtype, "attention on separate views vs area attention, 1080 keys")
        else:
            assert torch.equal(a, b), "area_attn (1080 keys) and attention on separate views differ"
    for heads, hd, (H, W), win, shift, pad in ((2, 16, (14, 18), 7, 0, False), (6, 8, (14, 18), 7, 3, True), (6, 8, (16, 20), 7, 3, True),
                                               (1, 16, (5, 4), 4, 0, False), (2, 32, (7, 7), 7, 0, False), (6, 8, (9, 11), 7, 0, True),
                                               # matrix-core kernel (16-bit, head_dim 16 / 32 / 64): rolled grid, pad vectors as keys, full 64-token windows
                                               (2, 32, (16, 20), 7, 3, True), (1, 64, (9, 11), 7, 0, True), (2, 16, (10, 12), 8, 4, True),
                                               (3, 32, (15, 22), 7, 3, False), (5, 16, (29, 31), 7, 0, True)):
        c = heads * hd
        qkv = _rnd(2, H, W, 3 * c, seed=50 + hd + shift, dtype=dtype)
        pads = [_rnd(c, seed=51 + i) if pad else None for i in range(3)]
        qd = qkv.to(DEV)
        got = ops.window_attention(qd[..., :c], qd[..., c:2 * c], qd[..., 2 * c:], heads, hd, hd ** -0.5, win, shift,
                                   *[p.to(DEV) if p is not None else None for p in pads])
        ref = emu_ops.window_attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], heads, hd, hd ** -0.5, win, shift, *pads)
        _cmp(got, ref, dtype, f"window h{heads} d{hd} {H}x{W} win{win} shift{shift} pad{pad}")
    for heads, hd, (H, W) in ((2, 16, (24, 28)), (1, 16, (20, 24)), (2, 32, (9, 9)), (1, 64, (10, 13))):
        c = heads * hd
        qkv = _rnd(2, H, W, 3 * c, seed=60 + hd, dtype=dtype)
        rf, _ = torch.linalg.qr(_rnd(hd, hd, seed=61))
        rf = rf[: min(64, hd)].contiguous()
        qd = qkv.to(DEV)
        _cmp(ops.linear_attention(qd[..., :c], qd[..., c:2 * c], qd[..., 2 * c:], rf.to(DEV), heads, hd),
             emu_ops.linear_attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], rf, heads, hd), dtype, f"linear h{heads} d{hd}")
    for heads, hd, (H, W), npnt, align in ((6, 8, (14, 18), 4, True), (8, 8, (15, 17), 4, True), (2, 16, (9, 1), 3, False), (1, 8, (1, 1), 4, True)):
        c = heads * hd
        v = _rnd(2, H, W, c, seed=70 + hd, dtype=dtype)
        off, aw = _rnd(2, H, W, heads * npnt * 2, seed=71, scale=1.5), _rnd(2, H, W, heads * npnt, seed=72)
        _cmp(ops.deform_attention(v.to(DEV), off.to(DEV), aw.to(DEV), heads, hd, npnt, align),
             emu_ops.deform_attention(v, off, aw, heads, hd, npnt, align), dtype, f"deform h{heads} d{hd} {H}x{W} np{npnt}")


# ------------------------------------------------------------------------------------------------- modules / model
def _load(golden_dir, fam, name):
    z = np.load(golden_dir / f"{fam}_{name}.npz")
    return z, {k: torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}


def _prep(mod, sd):
    mod.load_state_dict(sd)
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3
    return mod.eval().to(DEV)


@pytest.mark.parametrize("fam,name,ctor", [
    ("moa", "exact", ("MoABlock", (48,), dict(num_heads=6))), ("moa", "blend", ("MoABlock", (48,), dict(num_heads=6))),
    ("moa", "linear", ("MoABlock", (48,), dict(num_heads=6))),
    ("moa", "kvcap", ("MoABlock", (48,), dict(num_heads=6, regional_max_kv_tokens=64, shortcut=False))),
    ("moa", "c2f", ("C2fMoA", (64, 96), dict(n=2, num_heads=6))),
    ("moa", "hd21", ("MoABlock", (128,), dict(num_heads=6))),       # BASELINE config 5 (L scale): head_dim 21 padded to 24
    ("moa", "sparse", ("MoABlock", (48,), dict(num_heads=6, sparse_inference=True, sparse_inference_threshold=0.2))),       # a head group skipped for the batch
    ("moa", "sparse_one", ("MoABlock", (48,), dict(num_heads=6, sparse_inference=True, sparse_inference_threshold=0.99))),  # only the largest-mean group runs
    ("mot", "top2", ("MoTBlock", (48,), dict(num_heads=6))),
    ("mot", "shift", ("MoTBlock", (48,), dict(num_heads=6, window_shift=True, local_attn_window=7))),
    ("mot", "top1", ("MoTBlock", (48,), dict(num_heads=6, top_k=1))), ("mot", "dense", ("MoTBlock", (48,), dict(num_heads=6, top_k=3))),
    ("mot", "skip", ("MoTBlock", (48,), dict(num_heads=6))), ("mot", "c2f", ("C2fMoT", (64, 96), dict(n=2, num_heads=6))),
    # round 4: scene-aware residual / image-level router (mot/router.py:118-136, 166-240)
    ("mot", "scene", ("MoTBlock", (48,), dict(num_heads=6, scene_aware_router=True))),
    ("mot", "scene3", ("MoTBlock", (48,), dict(num_heads=6, scene_aware_router=True, scene_hidden_dim=5, top_k=1))),
    ("mot", "image", ("MoTBlock", (48,), dict(num_heads=6, use_spatial_router=False))),
    ("mot", "image_scene", ("MoTBlock", (48,), dict(num_heads=6, use_spatial_router=False, scene_aware_router=True))),
    ("mot", "scene_bypass", ("MoTBlock", (48,), dict(num_heads=6, scene_aware_router=True, scene_inference_mode="bypass"))),
    ("gated", "base", ("VisualEnhancedAdaptiveGateMoE", (64, 64), {})), ("gated", "small", ("VisualEnhancedAdaptiveGateMoE", (64, 64), {})),
    ("gated", "keep1", ("VisualEnhancedAdaptiveGateMoE", (64, 64), {})),
    ("gated", "e6k3", ("VisualEnhancedAdaptiveGateMoE", (96, 96), dict(num_experts=6, top_k=3))),
    ("gated", "e16", ("VisualEnhancedAdaptiveGateMoE", (64, 64), dict(num_experts=16, top_k=2))),
    ("gated", "mid", ("VisualEnhancedAdaptiveGateMoE", (64, 64), {})),
    # v0_1: ModularRouterExpertMoE (= OptimizedMOEImproved, moe/modules.py:957-1198), tests/golden/make_golden_v01.py
    ("v01", "base", ("ModularRouterExpertMoE", (64, 64, 4, 2), {})),
    ("v01", "e16", ("ModularRouterExpertMoE", (128, 128, 16, 2), {})),
    ("v01", "widen", ("ModularRouterExpertMoE", (64, 128, 8, 2), {})),
    ("v01", "small", ("ModularRouterExpertMoE", (64, 64, 4, 2), {})),
    ("v01", "k1", ("ModularRouterExpertMoE", (64, 64, 4, 1), {})),
    # v0_3: UltimateOptimizedMoE (moe/modules.py:1534-1700), tests/golden/make_golden_v03.py
    ("v03", "base", ("UltimateOptimizedMoE", (128, 128, 4, 2, 0.5), {})),
    ("v03", "e16", ("UltimateOptimizedMoE", (128, 128, 16, 2, 0.5), {})),
    ("v03", "lowc", ("UltimateOptimizedMoE", (128, 128, 8, 2, 0.5), {})),
    ("v03", "k1", ("UltimateOptimizedMoE", (128, 128, 4, 1, 0.5), {})),
    # v0_1 uomoe / exp v0_2: UltraOptimizedMoE (moe/modules.py:121-232), tests/golden/make_golden_uomoe.py
    ("uomoe", "base", ("UltraOptimizedMoE", (64, 64, 4, 2), {})),
    ("uomoe", "e16", ("UltraOptimizedMoE", (128, 128, 16, 2), {})),
    ("uomoe", "widen", ("UltraOptimizedMoE", (64, 128, 8, 2), {})),
    ("uomoe", "small", ("UltraOptimizedMoE", (64, 64, 4, 2), {})),
    ("uomoe", "thr", ("UltraOptimizedMoE", (64, 64, 4, 2), {})),      # a routed weight below the 0.01 inference threshold: dropped
])
def test_modules_vs_reference_golden(fam, name, ctor, golden_dir):
    """fp32 on the GPU against the REAL reference's outputs (the fixtures of tests/test_host_mixture.py)."""
    import warnings

    from yolo_master_amd.nn import mixture

    z, sd = _load(golden_dir, fam, name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = _prep(getattr(mixture, ctor[0])(*ctor[1], **ctor[2]), sd)
    with torch.inference_mode():
        got = m(torch.from_numpy(z["x"]).to(DEV)).cpu()
    ref = torch.from_numpy(z["y"])
    err = float((got - ref).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref.abs().max())), f"{fam}_{name}: max |d| {err:.3e}"
    if fam == "mot" and "scene_stats" in z.files:   # the scene statistics themselves (fp64 combines on the device against torch's fp32 reductions)
        st, rs = m.router.last_scene_stats.cpu(), torch.from_numpy(z["scene_stats"])[0]
        assert float(((st - rs).abs() / rs.abs().clamp_min(1e-3)).max()) <= 1e-4, f"scene statistics {st.tolist()} vs {rs.tolist()}"
    if fam in ("gated", "v01", "v03", "uomoe"):
        B = got.shape[0]
        assert np.array_equal(m.last_route["indices"].cpu().numpy(), z["indices"].reshape(B, -1)), "routed experts differ from the reference"
    if fam in ("v01", "v03", "uomoe"):
        zw = z["weights"] if fam != "uomoe" else np.where(z["weights"] > 0.01, z["weights"], 0.0)   # (the library hands back the weights as applied: <= 0.01 dropped)
        assert float(np.abs(m.last_route["weights"].cpu().numpy().reshape(B, -1) - zw).max()) <= 1e-5, "routing weights"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fam,name,ctor", [
    ("moa", "exact", ("MoABlock", (48,), dict(num_heads=6))), ("moa", "hd21", ("MoABlock", (128,), dict(num_heads=6))),
    ("moa", "kvcap", ("MoABlock", (48,), dict(num_heads=6, regional_max_kv_tokens=64, shortcut=False))),
    ("mot", "dense", ("MoTBlock", (48,), dict(num_heads=6, top_k=3))), ("mot", "top2", ("MoTBlock", (48,), dict(num_heads=6))),
    ("mot", "shift", ("MoTBlock", (48,), dict(num_heads=6, window_shift=True, local_attn_window=7))),
])
def test_modules_16bit_vs_reference_golden(fam, name, ctor, dtype, golden_dir):
    """The 16-bit forms of the MoA / MoT blocks (matrix-core attention, layer-scale factors folded into the convolution that feeds each
    residual — nn/mixture.py _fold_ls — which fp32 does not do) against the REAL reference's fp32 outputs.  Per-token routing can
    flip on near-ties in 16 bits (a flipped token differs by O(1)), so the bars are the 99th percentile and the mean of the error."""
    import warnings

    from yolo_master_amd import ops
    from yolo_master_amd.nn import mixture
    from yolo_master_amd.nn.modules import YmkModule

    if dtype == torch.float16 and not ops.HAS_F16:
        pytest.skip("libymk_f16.so not built")
    z, sd = _load(golden_dir, fam, name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = _prep(getattr(mixture, ctor[0])(*ctor[1], **ctor[2]), sd)
    for sub in m.modules():
        if isinstance(sub, YmkModule):
            sub.ymk_dtype = dtype
    with torch.inference_mode():
        got = m(torch.from_numpy(z["x"]).to(DEV)).float().cpu()
    ref = torch.from_numpy(z["y"])
    err = (got - ref).abs().reshape(-1)
    scale = max(1.0, float(ref.abs().max()))
    tol = (1e-2 if dtype == torch.bfloat16 else 2e-3) * scale
    q99, mean = float(torch.quantile(err, 0.99)), float(err.mean())
    print(f"{fam}_{name} {dtype}: |d| mean {mean:.2e} q99 {q99:.2e} max {float(err.max()):.2e} (scale {scale:.2f})")
    assert q99 <= tol and mean <= tol / 4, f"{fam}_{name} {dtype}: mean {mean:.3e}, q99 {q99:.3e} > {tol:.3e}"


@pytest.mark.parametrize("tag", ["cfg5", "v15", "v04", "v06", "v01", "v03", "v08s", "uomoe"])
def test_config5_model_vs_reference_golden(tag, golden_dir):
    import json
    import warnings

    from tests.helpers import fill_by_name
    from yolo_master_amd import ops
    from yolo_master_amd.nn.tasks import DetectionModel

    from tests.test_host_mixture import MODEL_FIXTURES

    z = np.load(golden_dir / f"fwd_{tag}.npz")
    cfg = json.loads(str(z["cfg"]))
    sd = fill_by_name(json.loads(str(z["spec"])), seed=5, gain=1.0)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DetectionModel(MODEL_FIXTURES[tag] or cfg)
    m.load_state_dict(sd)
    m.eval().to(DEV)
    taps = {}
    with torch.inference_mode():
        m._predict_once(torch.from_numpy(z["x"]).to(DEV), taps=taps)
    n = len(cfg["backbone"]) + len(cfg["head"])
    for i in range(n - 1):
        t = taps[i]
        if not torch.is_tensor(t):
            t = t.materialise()
        got = ops.nhwc_to_nchw_f32(t).cpu().reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        ref = z[f"layer{i}_val"]
        err = float(np.abs(got - ref).max() / max(1.0, float(np.abs(ref).max())))
        assert err <= 1e-3, f"layer {i} ({(cfg['backbone'] + cfg['head'])[i][2]}): scaled max error {err:.3e}"


@pytest.mark.parametrize("name", __import__("tests.test_host_mixture", fromlist=["GATED3_CASES"]).GATED3_CASES)
def test_gated_chain_vs_reference_golden(name, golden_dir):
    """AdaptiveGateMoE (v0_4) ... ContextRefinedLowRankHybridAdaptiveGateMoE, HybridAdaptiveGateMoEv2 (v0_11) (moe/gated.py:268-1700)
    on the GPU, fp32, against the real reference."""
    from tests.test_host_mixture import run_gated3_case

    run_gated3_case(name, golden_dir, dev=DEV, dtype=torch.float32, rtol=2e-4)


@pytest.mark.parametrize("name", __import__("tests.test_host_mixture", fromlist=["GATED2_CASES"]).GATED2_CASES)
def test_gated_v12_v15_vs_reference_golden(name, golden_dir):
    """OptimalHybridGateMoE / GatedFusionMoE (moe/gated.py:1846-2008, 2564-2693) on the GPU, fp32, against the real reference."""
    from tests.test_host_mixture import run_gated2_case

    run_gated2_case(name, golden_dir, dev=DEV, dtype=torch.float32, rtol=2e-4)


# ----------------------------------------------------------------------------- BASELINE configs[4] at its own configuration
def load_cfg5_l(golden_dir, setting="base"):
    """The L-scale 1280 x 1280 fixture of tests/golden/make_golden_cfg5_l.py: (npz, cfg dict, state_dict, x)."""
    import json

    from tests.helpers import cfg5_imbalance, condition_bn, fill_by_name
    from yolo_master_amd.weights import synth_input

    z = np.load(golden_dir / "fwd_cfg5_l.npz")
    cfg, rcp = json.loads(str(z["cfg"])), json.loads(str(z["recipe"]))
    sd = fill_by_name(json.loads(str(z["spec"])), seed=5, gain=1.0)
    condition_bn(sd)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    if setting == "imb":
        sd = cfg5_imbalance(sd, rcp["alpha_image"], rcp["alpha_token"])
    x = synth_input(rcp["batch"], rcp["img"], rcp["img"], seed=rcp["x_seed"])
    return z, cfg, rcp, sd, x


def _cfg5_module(m, name):
    mod = m
    for part in name.split("."):
        mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
    return mod


def check_cfg5_routes(m, z, tag):
    """Every discrete routing decision against the reference's: exact wherever the reference's own logit gap to the next expert is
    at least 1e-4 (the fixture lists the few per-token decisions below that as `close`: an evaluation-order difference of 1e-6 may
    resolve those ties either way).  Returns {MoT block name: bool mask [B, E, H, W] of the product's selection} for the blocks in
    which a close decision resolved differently (empty: identical routing everywhere)."""
    flipped = {}
    for key in [f for f in z.files if f.startswith(f"{tag}::route::")]:
        name = key[len(f"{tag}::route::"):]
        ref = z[key].astype(np.int64)
        r = _cfg5_module(m, name).last_route
        B, k = ref.shape[:2]
        if "indices" in r:                                # gated block: ranked experts per image
            got = r["indices"].cpu().numpy().reshape(ref.shape).astype(np.int64)
            assert np.array_equal(got, ref), f"{name}: routed experts differ from the reference"
            continue
        w = r["weights"].permute(0, 3, 1, 2).cpu()        # MoT block: selected experts per token
        sel = torch.zeros_like(w, dtype=torch.bool).scatter_(1, torch.from_numpy(ref), True)
        same = ((w > 0) == sel).all(1).reshape(B, -1).numpy()
        close = z[f"{tag}::close::{name}"]
        ok = same.copy()
        if len(close):
            ok[close[:, 0], close[:, 1]] = True
        assert ok.all(), f"{name}: {int((~ok).sum())} routing decisions differ from the reference outside its close calls"
        if not same.all():
            flipped[name] = w > 0
    return flipped


def oracle_under_the_products_tie_resolution(m, cfg, sd, x, names, margin):
    """The reference algorithm (oracle, bit-exact vs the real reference at this size) re-evaluated with the product's resolution of
    the near-ties: the first block (execution order) whose selection differs is forced to the product's, everything downstream is
    recomputed, and the comparison repeats until no selection differs; a difference is only accepted where the ORACLE's own logit
    gap is below `margin`.  Returns (y, taps) of that evaluation."""
    from oracle import model_ref, mot_ref

    mot_ref.FORCE_SELECT.clear()
    try:
        for _ in range(6):
            taps, info = {}, {}
            with torch.inference_mode():
                y, _, _ = model_ref.forward(cfg, sd, x, fused=False, taps=taps, moe_info=info)
            first = None
            for name in names:                            # execution order
                w = _cfg5_module(m, name).last_route["weights"].permute(0, 3, 1, 2).cpu() > 0
                ind = info[name]["indices"]
                sel = torch.zeros_like(w).scatter_(1, ind, True)
                diff = (w != sel).any(1)
                if not diff.any():
                    continue
                lg = info[name]["logits"].float()
                srt = lg.sort(dim=1, descending=True).values
                k = ind.shape[1]
                gap = srt[:, k - 1] - srt[:, k]
                assert float(gap[diff].max()) < margin, f"{name}: the product selects other experts where the reference's logit gap is {float(gap[diff].max()):.2e}"
                first = (name, w)
                break
            if first is None:
                return y, taps
            mot_ref.FORCE_SELECT[first[0] + ".router"] = first[1]
        raise AssertionError("tie resolution did not converge")
    finally:
        mot_ref.FORCE_SELECT.clear()


@pytest.mark.parametrize("setting", ["base", "imb"])
def test_config5_at_its_own_configuration(setting, golden_dir):
    """BASELINE.json configs[4] where it is defined: the v0_10 MoA + MoT detector at the L scale (51.7 M parameters), 2 x 3 x 1280 x
    1280, fp32, against the REAL reference (tests/golden/make_golden_cfg5_l.py): every layer and y within 1e-4 (layers: of the layer's
    scale; class scores: absolute; boxes: 1e-4 DFL bins = pixels / stride, as for config 2), routed experts per image and per token
    identical, NMS kept anchor indices and classes identical, Cluster-Weighted boxes (sigma 0.1, pinned to the reference's C++)
    within 1e-4 bins of the coarsest level.  `imb` = the expert-imbalance stress (all images / >= 90 % of the tokens on expert 0).

    Per-token routing has ~150 000 top-k decisions per forward; a handful have logit gaps of 1e-6 ... 1e-8 (the fixture lists those
    under 1e-4), which two correct fp32 evaluations may resolve differently.  When that happens the comparison switches from the
    committed vectors to the oracle re-evaluated under the product's resolution of exactly those ties (full tensors, same bars)."""
    import warnings

    from oracle import nms_ref
    from yolo_master_amd.nms import non_max_suppression
    from yolo_master_amd.nn.tasks import DetectionModel

    z, cfg, rcp, sd, x = load_cfg5_l(golden_dir, setting)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DetectionModel(cfg)
    m.load_state_dict(sd)
    m.eval().to(DEV)
    taps = {}
    with torch.inference_mode():
        y, _ = m._predict_once(x.to(DEV), taps=taps)
    m.check_flags()
    flipped = check_cfg5_routes(m, z, setting)
    rows = cfg["backbone"] + cfg["head"]
    n = len(rows)
    B, ch, A = y.shape
    img = rcp["img"]
    stride_of = torch.cat([torch.full(((img // s) ** 2,), float(s)) for s in (8, 16, 32)])
    worst = 0.0
    if not flipped:                                       # the committed vectors of the real reference
        if setting == "base":
            for i in range(n - 1):
                t = taps[i] if torch.is_tensor(taps[i]) else taps[i].materialise()
                got = ops_nchw(t).reshape(-1)[torch.from_numpy(z[f"base::layer{i}_idx"])].numpy()
                ref = z[f"base::layer{i}_val"]
                err = float(np.abs(got - ref).max() / max(1.0, float(np.abs(ref).max())))
                worst = max(worst, err)
                assert err <= 1e-4, f"layer {i} ({rows[i][2]}): scaled max error {err:.3e}"
        yi = torch.from_numpy(z[f"{setting}::y_idx"])
        got, ref = y.cpu().reshape(-1)[yi], torch.from_numpy(z[f"{setting}::y_val"])
        row, anchor = (yi // A) % ch, yi % A
        e_box = float(((got - ref).abs() / stride_of[anchor])[row < 4].max())
        e_cls = float((got - ref).abs()[row >= 4].max())
        ref_nms = [(z[f"{setting}::nms_det{b}"], z[f"{setting}::nms_idx{b}"], z[f"{setting}::cw_box{b}"]) for b in range(B)]
    else:                                                 # the reference algorithm under the product's resolution of the listed ties
        names = [k[len(f"{setting}::route::"):] for k in z.files if k.startswith(f"{setting}::route::") and ".m." in k]
        oy, otaps = oracle_under_the_products_tie_resolution(m, cfg, sd, x, names, rcp["margin"])
        for i in range(n - 1):
            t = taps[i] if torch.is_tensor(taps[i]) else taps[i].materialise()
            err = float((ops_nchw(t) - otaps[i]).abs().max() / max(1.0, float(otaps[i].abs().max())))
            worst = max(worst, err)
            assert err <= 1e-4, f"layer {i} ({rows[i][2]}): scaled max error {err:.3e} (full tensor, ties resolved as the product did)"
        d = (y.cpu() - oy).abs()
        e_box, e_cls = float((d[:, :4] / stride_of).max()), float(d[:, 4:].max())
        dets_o, idx_o = nms_ref.non_max_suppression(oy.numpy(), rcp["conf"], rcp["iou"], return_idxs=True)
        ref_nms = []
        for b in range(B):
            yb = oy[b].numpy()
            conf, cls = yb[4:].max(0), yb[4:].argmax(0)
            mk = conf > np.float32(rcp["conf"])
            cands = np.concatenate([nms_ref.xywh2xyxy(yb[:4].T.copy())[mk], conf[mk, None], cls[mk, None].astype(np.float32)], 1).astype(np.float32)
            pos = {int(a): j for j, a in enumerate(np.arange(A)[mk])}
            keep = np.array([pos[int(a)] for a in idx_o[b]], np.int64)
            ref_nms.append((dets_o[b], idx_o[b], nms_ref.cw_refine(cands, keep, rcp["iou"], rcp["sigma"])))
    print(f"config 5 @ L, {img}^2 [{setting}]: worst layer {worst:.2e}, boxes {e_box:.2e} bins, scores {e_cls:.2e}; "
          f"near-ties resolved differently in {sorted(flipped)}")
    assert e_box <= 1e-4 and e_cls <= 1e-4, (e_box, e_cls)
    dets, idx = non_max_suppression(y, rcp["conf"], rcp["iou"], return_idxs=True)
    cw, _ = non_max_suppression(y, rcp["conf"], rcp["iou"], return_idxs=True, cluster=True, sigma=rcp["sigma"])
    for b in range(B):
        ref_det, ref_idx, ref_cw = ref_nms[b]
        assert np.array_equal(idx[b].cpu().numpy(), ref_idx), f"image {b}: NMS kept anchors differ"
        d = dets[b].cpu().numpy()
        assert np.array_equal(d[:, 5], ref_det[:, 5]), f"image {b}: classes differ"
        s = stride_of[torch.from_numpy(np.asarray(ref_idx))].numpy()
        assert float((np.abs(d[:, :4] - ref_det[:, :4]).max(1) / s).max()) <= 1e-4 and float(np.abs(d[:, 4] - ref_det[:, 4]).max()) <= 1e-4
        c = cw[b].cpu().numpy()
        assert np.array_equal(c[:, 4:], d[:, 4:]), "Cluster-Weighted refinement must leave survivors, scores and classes untouched"
        e_cw = float(np.abs(c[:, :4] - ref_cw).max())
        assert e_cw <= 32 * 1e-4, f"image {b}: Cluster-Weighted boxes off by {e_cw:.3e} px"
        assert float(np.abs(c[:, :4] - d[:, :4]).max()) > 1.0, "the fixture's clusters move boxes by tens of pixels"


def test_config5_fp16_vs_the_references_own_half_run(golden_dir):
    """BASELINE.json configs[4] at its stated precision — "fp16 + CW-NMS": the L-scale MoA + MoT detector at 2 x 1280^2 on libymk_f16
    with Cluster-Weighted NMS, held to the REFERENCE'S OWN fp16 evaluation (tests/golden/make_golden_cfg5_l_ref16.py: the real reference
    `.half()` on the same weights and images, measured against its own fp32 result — the method of config 3).  libymk's fp16 result must
    be at least as close to the reference's fp32 result as the reference's fp16 run is: score / box error percentiles (x 1.25: the two
    evaluations round at different points of the graph), routed experts per image identical, per-token selections agreeing at least as
    often (- 0.3 %), NMS kept-set Jaccard (- 0.03) and the Cluster-Weighted boxes of the common survivors."""
    import json
    import warnings

    from yolo_master_amd import ops
    from yolo_master_amd.nms import non_max_suppression
    from yolo_master_amd.nn.tasks import DetectionModel

    if not ops.HAS_F16:
        pytest.skip("libymk_f16.so not built")
    z, cfg, rcp, sd, x = load_cfg5_l(golden_dir, "base")
    r16 = np.load(golden_dir / "cfg5_l_ref16.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DetectionModel(cfg)
    m.load_state_dict(sd)
    m.eval().to(DEV).set_compute_dtype(torch.float16)
    with torch.inference_mode():
        y, _ = m._predict_once(x.to(DEV))
    m.check_flags()
    B, ch, A = y.shape
    img = rcp["img"]
    stride_of = torch.cat([torch.full(((img // s) ** 2,), float(s)) for s in (8, 16, 32)])
    yi = torch.from_numpy(z["base::y_idx"])
    got, ref = y.cpu().reshape(-1)[yi], torch.from_numpy(z["base::y_val"])
    row = (yi // A) % ch
    d = (got - ref).abs()
    pct = tuple(json.loads(str(r16["recipe"]))["percentiles"])
    sc, bx = np.percentile(d[row >= 4].numpy(), pct), np.percentile(d[row < 4].numpy(), pct)
    print(f"config 5 fp16 vs the reference's fp32 (libymk | the reference's own fp16 run), percentiles {pct}: scores {sc} | {r16['score_err_pct']}; "
          f"boxes px {bx} | {r16['box_err_px_pct']}")
    assert (sc <= 1.25 * r16["score_err_pct"] + 1e-6).all() and (bx <= 1.25 * r16["box_err_px_pct"] + 1e-3).all()
    agree = json.loads(str(r16["route_agreement"]))
    for key in [f for f in z.files if f.startswith("base::route::")]:
        name = key[len("base::route::"):]
        refi = z[key].astype(np.int64)
        r = _cfg5_module(m, name).last_route
        if "indices" in r:
            a = float((r["indices"].cpu().numpy().reshape(refi.shape) == refi).all(axis=tuple(range(1, refi.ndim))).mean())
            want = agree.get(name + ".routing", 1.0)
        else:
            w = r["weights"].permute(0, 3, 1, 2).cpu()
            sel = torch.zeros_like(w, dtype=torch.bool).scatter_(1, torch.from_numpy(refi), True)
            a = float(((w > 0) == sel).all(1).float().mean())
            want = agree.get(name + ".router", 1.0)
        assert a >= want - 3e-3, f"{name}: routing agrees with the fp32 reference on {a:.4f} of the decisions; the reference's own fp16 run on {want:.4f}"
    dets, idx = non_max_suppression(y, rcp["conf"], rcp["iou"], return_idxs=True)
    cw, _ = non_max_suppression(y, rcp["conf"], rcp["iou"], return_idxs=True, cluster=True, sigma=rcp["sigma"])
    cwd = []
    for b in range(B):
        ref_idx, ref_cw = z[f"base::nms_idx{b}"], z[f"base::cw_box{b}"]
        g = idx[b].cpu().numpy()
        s32, s16 = set(ref_idx.tolist()), set(g.tolist())
        jac = len(s32 & s16) / max(len(s32 | s16), 1)
        assert jac >= float(r16["nms_jaccard"][b]) - 0.03, f"image {b}: kept-set Jaccard {jac:.3f} vs the reference's own fp16 run {float(r16['nms_jaccard'][b]):.3f}"
        c = cw[b].cpu().numpy()
        assert np.array_equal(c[:, 4:], dets[b].cpu().numpy()[:, 4:])
        pos = {int(a_): j for j, a_ in enumerate(g)}
        common = [(j, pos[int(a_)]) for j, a_ in enumerate(ref_idx) if int(a_) in pos]
        ja, jb = np.array(common).T
        cwd.append(np.abs(ref_cw[ja] - c[jb, :4]).max(1))
    cwp = np.percentile(np.concatenate(cwd), pct)
    print(f"config 5 fp16 + CW-NMS: Cluster-Weighted boxes of the common survivors vs the fp32 reference, px percentiles {pct}: {cwp} | {r16['cw_box_err_px_pct']}")
    assert (cwp[:2] <= 1.25 * r16["cw_box_err_px_pct"][:2] + 1e-2).all()      # (p99 of ~300 boxes is a single cluster: reported, not held)


def ops_nchw(t):
    from yolo_master_amd import ops

    return ops.nhwc_to_nchw_f32(t).cpu()
