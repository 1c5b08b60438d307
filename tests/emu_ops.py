"""TEST INFRASTRUCTURE — a torch-CPU stand-in for the libymk entry points, one function per wrapper in
``yolo_master_amd/ops.py``, written from the contracts in ``include/ymk.h`` (argument layout, packed weight formats,
what each output holds).  It lets the ``-m "not gpu"`` suite drive the product's HOST code (weight packing, channel
permutations, buffer slicing, graph walk, routing bookkeeping) end to end and compare the result with the oracle.

It is installed by the ``emu`` fixture (monkeypatching ``yolo_master_amd.ops``) and exists only under ``tests/``: the
product never imports it and still refuses CPU tensors (``test_host_logic.py::test_no_cpu_fallback``).  Arithmetic is
fp32 on the host with the output rounded to the tensor's dtype, i.e. what a kernel with fp32 accumulation produces.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from oracle import nms_ref

CALLS: dict[str, int] = {}


def _count(name):
    CALLS[name] = CALLS.get(name, 0) + 1


def _nchw(x):
    return x.float().permute(0, 3, 1, 2)


def _finish(y_nchw, act, residual, out, dtype):
    if act:
        y_nchw = F.silu(y_nchw)
    y = y_nchw.permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual.float()
    if out is None:
        return y.to(dtype).contiguous()
    assert tuple(out.shape) == tuple(y.shape), f"out {tuple(out.shape)} vs result {tuple(y.shape)}"
    out.copy_(y.to(out.dtype))
    return out


def _unpack_conv(w_packed, k, cin):
    cout = w_packed.shape[0]
    assert w_packed.shape[1] % 64 == 0 and w_packed.shape[1] >= k * k * cin, "packed rows are Kpad (multiple of 64) long"
    assert float(w_packed[:, k * k * cin:].abs().sum()) == 0.0, "K padding must be zero"
    return w_packed[:, : k * k * cin].float().reshape(cout, k, k, cin).permute(0, 3, 1, 2)


def conv2d(x, w_packed, bias, k, stride, act, out=None, residual=None, out_dtype=None):
    _count("conv2d")
    assert w_packed.dtype == x.dtype and bias.dtype == torch.float32
    w = _unpack_conv(w_packed, k, x.shape[-1])
    y = F.conv2d(_nchw(x), w, bias, stride, k // 2)
    return _finish(y, act, residual, out, out_dtype or x.dtype)


def conv1x1_cat2(x1, up1, x2, w_packed, bias, act, out=None):
    _count("conv1x1_cat2")
    if up1:
        x1 = x1.repeat_interleave(2, 1).repeat_interleave(2, 2)
    return conv2d(torch.cat([x1, x2], -1), w_packed, bias, 1, 1, act, out=out)


def conv2d_stem(x_nchw, w, bias, k, stride, act, dtype, out=None, wt=None):
    _count("conv2d_stem")
    cout, cin = w.shape[0], x_nchw.shape[1]
    assert w.dtype == torch.float32 and w.shape[1] == k * k * cin
    if wt is not None:
        assert torch.equal(wt, w.t())
    y = F.conv2d(x_nchw.float(), w.reshape(cout, k, k, cin).permute(0, 3, 1, 2), bias, stride, k // 2)
    return _finish(y, act, None, out, dtype)


def _dw(x_nchw, w_kkc, k):
    c = x_nchw.shape[1]
    assert tuple(w_kkc.shape) == (k * k, c)
    return F.conv2d(x_nchw, w_kkc.float().t().reshape(c, 1, k, k), None, 1, k // 2, 1, c)


def dwconv2d(x, w_packed, bias, k, act, out=None, residual=None):
    _count("dwconv2d")
    y = _dw(_nchw(x), w_packed, k)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return _finish(y, act, residual, out, x.dtype)


def dwpw_supported(dtype, C, kmax):
    return False


def dwconv_pwconv(x, dw_w, dw_b, k, dw_act, pw_w, pw_b, pw_act, out=None):
    _count("dwconv_pwconv")
    h = dwconv2d(x, dw_w, dw_b, k, dw_act)
    return conv2d(h, pw_w, pw_b, 1, 1, pw_act, out=out)


# ------------------------------------------------------------------------------------------------- ES-MoE
def esmoe_route(x, w1, b1, w2, b2, top_k, thr, flags):
    """include/ymk.h `ymk_esmoe_route`: GAP -> MLP -> clamped softmax -> hard top-k -> retained set + CSR."""
    _count("esmoe_route")
    B = x.shape[0]
    E = w2.shape[0]
    pooled = x.float().mean((1, 2))
    if not bool(torch.isfinite(pooled).all()):
        flags |= 1
    logits = F.silu(pooled @ w1.t() + b1) @ w2.t() + b2
    if not bool(torch.isfinite(logits).all()):
        flags |= 2
    w = F.softmax(logits.clamp(-30.0, 30.0), 1)
    vals, idx = torch.topk(w, top_k, 1)
    vals = vals / vals.sum(1, keepdim=True).clamp_min(1e-6)
    route_w = torch.zeros_like(w).scatter_(1, idx, vals)
    if top_k >= E:
        retained = torch.ones(B, E, dtype=torch.bool)
        gate_w = route_w.clone()
    else:
        topv, topi = torch.topk(route_w, top_k, 1)
        keep = torch.ones_like(topi, dtype=torch.bool)
        if thr > 0:
            keep = (torch.arange(top_k).view(1, -1) == 0) | (topv >= thr)
        retained = torch.zeros(B, E, dtype=torch.bool).scatter_(1, topi, keep)
        rw = route_w * retained
        gate_w = rw / rw.sum(1, keepdim=True).clamp_min(torch.finfo(torch.float32).eps)
    sel = torch.full((B, top_k), -1, dtype=torch.int32)
    pairs = [[] for _ in range(E)]
    for b in range(B):
        for slot, e in enumerate(torch.where(retained[b])[0].tolist()):   # ascending expert order
            sel[b, slot] = e
            pairs[e].append(b * top_k + slot)
    csr_off = torch.zeros(E + 1, dtype=torch.int32)
    csr_pair = torch.zeros(B * top_k, dtype=torch.int32)
    n = 0
    for e in range(E):
        csr_off[e] = n
        for p in pairs[e]:
            csr_pair[n] = p
            n += 1
    csr_off[E] = n
    return route_w, gate_w, sel, csr_off, csr_pair


def esmoe_dw(x, dw_w, dw_off, ksizes, kmax, top_k, sel, csr_off, csr_pair):
    _count("esmoe_dw")
    B, H, W, C = x.shape
    assert kmax == int(ksizes.max())
    out = torch.zeros((B * top_k, H, W, C), dtype=x.dtype)
    xn = _nchw(x)
    E = ksizes.numel()
    for e in range(E):   # walk the CSR exactly as a kernel would
        k = int(ksizes[e])
        w = dw_w[int(dw_off[e]): int(dw_off[e]) + k * k * C].reshape(k * k, C)
        for n in range(int(csr_off[e]), int(csr_off[e + 1])):
            pair = int(csr_pair[n])
            b, slot = divmod(pair, top_k)
            assert int(sel[b, slot]) == e
            out[pair] = _dw(xn[b: b + 1], w, k)[0].permute(1, 2, 0).to(x.dtype)
    return out


def esmoe_pw(dw_out, B, H, W, pw_w, pw_b, nscale, nshift, top_k, sel, gate_w, out=None):
    _count("esmoe_pw")
    C = dw_out.shape[-1]
    E, Cout, Kp = pw_w.shape
    y = torch.zeros((B, H, W, Cout), dtype=torch.float32)
    for b in range(B):
        for slot in range(top_k):
            e = int(sel[b, slot])
            if e < 0:
                continue
            h = dw_out[b * top_k + slot].float() @ pw_w[e, :, :C].float().t() + pw_b[e]
            y[b] += F.silu(h) * gate_w[b, e]
    y = F.silu(y * nscale + nshift)
    if out is None:
        return y.to(dw_out.dtype)
    out.copy_(y.to(out.dtype))
    return out


def esmoe_experts_fused(x, dw_w, dw_off, ksizes, kmax, pw_w, pw_b, nscale, nshift, top_k, sel, gate_w, out=None):
    raise AssertionError("host code must ask dwpw_supported() first")


# ------------------------------------------------------------------------------------------------- attention / layout
def area_attn(qkv, heads, area, out=None):
    """`ymk_area_attn`: channels [Q | K | V], each [heads][32]; `area` contiguous token ranges."""
    _count("area_attn")
    B, H, W, C3 = qkv.shape
    d, N = 32, H * W
    cq = heads * d
    assert C3 == 3 * cq and N % area == 0
    t = qkv.float().reshape(B * area, N // area, 3, heads, d)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))          # [B*area, heads, n, d]
    p = torch.softmax((q @ k.transpose(-1, -2)) * d ** -0.5, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, H, W, cq)
    if out is None:
        return o.to(qkv.dtype)
    out.copy_(o.to(out.dtype))
    return out


def upsample2x(x, out=None):
    _count("upsample2x")
    y = x.repeat_interleave(2, 1).repeat_interleave(2, 2)
    if out is None:
        return y.contiguous()
    out.copy_(y)
    return out


def copy_channels(x, out):
    _count("copy_channels")
    out.copy_(x)
    return out


def scale_residual(y, gamma, residual, out=None):
    _count("scale_residual")
    r = residual.float() + gamma.float() * y.float()
    if out is None:
        return r.to(y.dtype)
    out.copy_(r.to(out.dtype))
    return out


def nhwc_to_nchw_f32(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------- detect / nms
def detect_decode(box_l, cls_l, y, stride, a_off, reg_max):
    _count("detect_decode")
    B, Hl, Wl, _ = box_l.shape
    nc = cls_l.shape[-1]
    n = Hl * Wl
    dist = (box_l.reshape(B, n, 4, reg_max).softmax(-1) * torch.arange(reg_max, dtype=torch.float32)).sum(-1)
    sy, sx = torch.meshgrid(torch.arange(Hl, dtype=torch.float32) + 0.5, torch.arange(Wl, dtype=torch.float32) + 0.5,
                            indexing="ij")
    anc = torch.stack((sx, sy), -1).view(1, n, 2)
    x1y1, x2y2 = anc - dist[..., :2], anc + dist[..., 2:]
    box = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1) * stride
    y[:, :4, a_off: a_off + n] = box.transpose(1, 2)
    y[:, 4:, a_off: a_off + n] = cls_l.reshape(B, n, nc).sigmoid().transpose(1, 2)
    return y


def nms_batched(y, conf, iou, multi_label, agnostic, max_det, max_nms, max_wh, cw_sigma=None, cw_pool=3000):
    _count("nms_batched")
    assert cw_sigma is None, "CW refinement is checked on the GPU against oracle/_ref"
    B = y.shape[0]
    outs, idxs = nms_ref.non_max_suppression(y.numpy(), conf, iou, multi_label, agnostic, max_det, max_nms, max_wh,
                                             return_idxs=True)
    dets = torch.zeros((B, max_det, 6), dtype=torch.float32)
    counts = torch.zeros((B,), dtype=torch.int32)
    idx = torch.zeros((B, max_det), dtype=torch.int32)
    for b in range(B):
        n = len(outs[b])
        counts[b] = n
        dets[b, :n] = torch.from_numpy(np.asarray(outs[b], np.float32))
        idx[b, :n] = torch.from_numpy(np.asarray(idxs[b]).astype(np.int32))
    return dets, counts, idx, torch.zeros((1,), dtype=torch.int32)


EMULATED = ["conv2d", "conv1x1_cat2", "conv2d_stem", "dwconv2d", "dwpw_supported", "dwconv_pwconv", "esmoe_route", "esmoe_dw",
            "esmoe_pw", "esmoe_experts_fused", "area_attn", "upsample2x", "copy_channels", "scale_residual", "nhwc_to_nchw_f32",
            "detect_decode", "nms_batched"]
