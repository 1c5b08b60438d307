"""Seeded random-shape sweep of the config-5 entry points on the CPU lane emulator (tests/hostemu): odd sizes, single
pixels, single images, strided views, every dtype — against the contract restatements of tests/emu_ops.py.  The fixed
cases of tests/test_gpu_mixture.py pin the shapes the models use; this sweep looks for indexing mistakes at the edges."""
import random

import pytest
import torch

from tests import emu_ops

TOL = {torch.float32: 3e-5, torch.bfloat16: 1.6e-2}


def _cmp(got, ref, dtype, what):
    err = float((got.float() - ref.float()).abs().max())
    assert err <= TOL[dtype] * max(1.0, float(ref.float().abs().max())), f"{what}: max |d| {err:.3e}"


def _t(rng, shape, dtype, pad=0, scale=1.0):
    g = torch.Generator().manual_seed(rng.randrange(1 << 30))
    t = (torch.randn(*shape, generator=g) * scale).to(dtype)
    if not pad:
        return t
    wide = torch.zeros((*shape[:3], shape[3] + pad), dtype=dtype)
    wide[..., : shape[3]] = t
    return wide[..., : shape[3]]


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes(T, seed):
    from yolo_master_amd import ops

    rng = random.Random(1000 + seed)
    dtype = rng.choice([torch.float32, torch.bfloat16])
    vec = 8 if dtype == torch.bfloat16 else 4
    B, H, W = rng.randint(1, 3), rng.randint(1, 9), rng.randint(1, 9)
    C = rng.choice([vec, 2 * vec, 3 * vec, 6, 10, 5 * vec])
    pad = rng.choice([0, vec, 2 * vec])
    x, y = _t(rng, (B, H, W, C), dtype, pad), _t(rng, (B, H, W, C), dtype, rng.choice([0, vec]))
    tag = f"seed {seed}: {dtype} B{B} {H}x{W} C{C} pad{pad}"
    w, b = 1 + 0.1 * _t(rng, (C,), torch.float32), 0.1 * _t(rng, (C,), torch.float32)
    groups = rng.choice([g for g in (1, 2, 3, 4, 5, 8) if C % g == 0])
    act = rng.choice([False, "silu"])
    res = rng.choice([None, y])
    _cmp(ops.group_norm(x, groups, w, b, 1e-5, act=act, residual=res), emu_ops.group_norm(x, groups, w, b, 1e-5, act=act, residual=res), dtype,
         f"group_norm g{groups} {tag}")
    _cmp(ops.layer_norm(x, w, b, 1e-5), emu_ops.layer_norm(x, w, b, 1e-5), dtype, f"layer_norm {tag}")
    aa = rng.choice([None, "sigmoid"])
    _cmp(ops.eltwise_mul(x, y, act_a=aa), emu_ops.eltwise_mul(x, y, act_a=aa), dtype, f"mul {tag}")
    al = rng.random()
    _cmp(ops.lerp(x, y, al), emu_ops.lerp(x, y, al), dtype, f"lerp {tag}")
    gate = torch.sigmoid(_t(rng, (B, 1, 1, C), torch.float32))
    _cmp(ops.channel_gate(x, gate), emu_ops.channel_gate(x, gate), dtype, f"channel_gate {tag}")
    if (H, W) != (1, 1):
        _cmp(ops.fma_gate(x, y, gate, 0.3), emu_ops.fma_gate(x, y, gate, 0.3), dtype, f"fma_gate image {tag}")
    _cmp(ops.fma_gate(x, x, y, -0.4), emu_ops.fma_gate(x, x, y, -0.4), dtype, f"fma_gate map {tag}")
    E = rng.randint(1, 4)
    parts = [_t(rng, (B, H, W, C), dtype) for _ in range(E)]
    wts = torch.softmax(_t(rng, (B, H, W, E), torch.float32), -1)
    _cmp(ops.weighted_sum(wts, parts), emu_ops.weighted_sum(wts, parts), dtype, f"weighted_sum E{E} {tag}")
    Ho, Wo = rng.randint(1, H), rng.randint(1, W)
    _cmp(ops.adaptive_avg_pool(x, Ho, Wo), emu_ops.adaptive_avg_pool(x, Ho, Wo), dtype, f"adaptive {Ho}x{Wo} {tag}")
    k = rng.randint(1, min(H, W))
    _cmp(ops.avg_pool(x, k, out_dtype=torch.float32), emu_ops.avg_pool(x, k, out_dtype=torch.float32), torch.float32, f"avg_pool {k} {tag}")
    ws = rng.random() < 0.5
    _cmp(ops.channel_stats(x, want_std=ws), emu_ops.channel_stats(x, want_std=ws), torch.float32, f"stats {tag}")
    small = [_t(rng, (B, max(1, H // s), max(1, W // s), C), dtype) for s in (2, 4)]
    _cmp(ops.mean_upsampled([x] + small), emu_ops.mean_upsampled([x] + small), dtype, f"mean_upsampled {tag}")
    g2 = rng.choice([g for g in (1, 2, 4) if (2 * C) % g == 0])
    assert torch.equal(ops.channel_shuffle_cat([x, y], g2), emu_ops.channel_shuffle_cat([x, y], g2)), f"shuffle g{g2} {tag}"
    n = rng.randint(2, 4)
    tk = rng.choice([0, rng.randint(1, n - 1)])
    lg = _t(rng, (B, H, W, 4), torch.float32, scale=2.0)
    wv, av = ops.token_softmax(lg, n, 1.3, top_k=tk)
    rw, ra = emu_ops.token_softmax(lg, n, 1.3, top_k=tk)
    assert torch.equal(wv > 0, rw > 0) and torch.equal(av, ra), f"token_softmax n{n} k{tk} {tag}"
    _cmp(wv, rw, torch.float32, f"token_softmax {tag}")


@pytest.mark.parametrize("seed", range(16))
def test_random_attention_shapes(T, seed):
    from yolo_master_amd import ops

    rng = random.Random(2000 + seed)
    dtype = rng.choice([torch.float32, torch.bfloat16])
    B, H, W = rng.randint(1, 2), rng.randint(1, 11), rng.randint(1, 11)
    heads, hd = rng.randint(1, 3), rng.choice([8, 16, 24, 32])
    c = heads * hd
    tag = f"seed {seed}: {dtype} B{B} {H}x{W} h{heads} d{hd}"
    qkv = _t(rng, (B, H, W, 3 * c), dtype)
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    Hk, Wk = rng.randint(1, H), rng.randint(1, W)
    kv = _t(rng, (B, Hk, Wk, 2 * c), dtype)
    _cmp(ops.attention(q, kv[..., :c], kv[..., c:], heads, hd, hd ** -0.5), emu_ops.attention(q, kv[..., :c], kv[..., c:], heads, hd, hd ** -0.5),
         dtype, f"attention {Hk}x{Wk} {tag}")
    win = rng.randint(1, 7)
    shift = rng.choice([0, win // 2])
    pads = [_t(rng, (c,), torch.float32) if rng.random() < 0.5 else None for _ in range(3)]
    _cmp(ops.window_attention(q, k, v, heads, hd, 0.3, win, shift, *pads), emu_ops.window_attention(q, k, v, heads, hd, 0.3, win, shift, *pads), dtype,
         f"window win{win} shift{shift} {tag}")
    nb = rng.randint(1, hd)
    rf = _t(rng, (nb, hd), torch.float32, scale=hd ** -0.5).contiguous()
    _cmp(ops.linear_attention(q, k, v, rf, heads, hd), emu_ops.linear_attention(q, k, v, rf, heads, hd), dtype, f"linear nb{nb} {tag}")
    npnt = rng.randint(1, 5)
    off, aw = _t(rng, (B, H, W, heads * npnt * 2), torch.float32, scale=2.0), _t(rng, (B, H, W, heads * npnt), torch.float32)
    al = rng.random() < 0.5
    _cmp(ops.deform_attention(v.contiguous(), off, aw, heads, hd, npnt, al), emu_ops.deform_attention(v, off, aw, heads, hd, npnt, al), dtype,
         f"deform np{npnt} align{al} {tag}")
