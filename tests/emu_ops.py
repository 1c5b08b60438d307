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
    if isinstance(act, str) and act != "silu":      # "gelu" / "sigmoid": the extended epilogues of ymk_conv2d (ymk.h YMK_ACT_*)
        y_nchw = _act(y_nchw, act)
    elif act:
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


def _ld(t):
    """Pixel stride of an NHWC view under the rules of the product's wrappers (ops._nhwc): channel-dense, pixels and
    images contiguous."""
    B, H, W, C = t.shape
    assert t.stride(3) == 1 or C == 1, "NHWC view must be channel-dense"
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else max(C, t.stride(0) // max(H * W, 1)))
    assert not (W > 1 and H > 1) or t.stride(1) == W * ld, "rows must be contiguous in pixels"
    assert B == 1 or t.stride(0) == H * W * ld, "images must be contiguous in pixels"
    return ld


def _vec(dtype):
    return 8 if dtype == torch.bfloat16 else 4


def _rules(x, cout=None, out=None, residual=None):
    """The argument rules libymk enforces (YMK_E_BADARG in csrc/*.hip): 16-byte channel vectors on the input side,
    4-element granularity on the output side."""
    v = _vec(x.dtype)
    assert x.shape[-1] % v == 0 and _ld(x) % v == 0, f"input channels {x.shape[-1]} / stride {_ld(x)} not a multiple of {v}"
    if cout is not None:
        assert cout % 4 == 0, f"Cout {cout} not a multiple of 4"
    if out is not None:
        assert _ld(out) % 4 == 0
    if residual is not None:
        assert _ld(residual) % 4 == 0


def conv2d(x, w_packed, bias, k, stride, act, out=None, residual=None, out_dtype=None, pool=False):
    _count("conv2d")   # (pool: the library may leave `out.gap_part`; the emulated router always pools the map itself)
    assert w_packed.dtype == x.dtype and bias.dtype == torch.float32
    assert k in (1, 3) and stride in (1, 2)
    _rules(x, w_packed.shape[0], out, residual)
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
    assert k % 2 == 1 and k <= 15 and w_packed.dtype == x.dtype
    _rules(x, None, out, residual)
    y = _dw(_nchw(x), w_packed, k)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return _finish(y, act, residual, out, x.dtype)


def stem_pair_supported(dtype, cin, c0, c1, k0, s0, k1, s1):
    return dtype == torch.bfloat16 and (cin, c0, c1, k0, s0, k1, s1) == (3, 32, 64, 3, 2, 3, 2)


def stem_pair(x_nchw, wt0, b0, w1, b1, out=None):
    """include/ymk.h `ymk_stem_pair`: the stem (fp32 operands, bf16 map) followed by the next 3x3/s2 convolution."""
    _count("stem_pair")
    h = conv2d_stem(x_nchw, wt0.t().contiguous(), b0, 3, 2, True, torch.bfloat16)
    return conv2d(h, w1, b1, 3, 2, True, out=out)


def c3k2_fused_supported(dtype, c1, c2, c, n, c3k, shortcut):
    return dtype == torch.bfloat16 and (c1, c2, c, n) == (64, 128, 32, 1) and not c3k and bool(shortcut)


def c3k2_fused(x, p1, pa, pb, p2, out=None, pool=True):
    """include/ymk.h `ymk_c3k2_fused`: cv1 -> Bottleneck(3x3, 3x3, + residual) -> cv2 over [a | b | m], each stage rounded to bf16."""
    _count("c3k2_fused")
    y1 = conv2d(x, p1[0], p1[1], 1, 1, True)
    b = y1[..., 32:]
    h = conv2d(b, pa[0], pa[1], 3, 1, True)
    m = conv2d(h, pb[0], pb[1], 3, 1, True, residual=b)
    return conv2d(torch.cat([y1, m], -1), p2[0], p2[1], 1, 1, True, out=out)


def detect_cls_fused_supported(dtype, cin, c3, nc):
    return dtype == torch.bfloat16 and cin in (128, 256) and c3 == 128 and 1 <= nc <= 128


def detect_cls_fused(x, d1, p1, d2, p2, w3, out=None, y=None, nc=0, a_off=0, raw=True):
    """include/ymk.h `ymk_detect_cls_fused`: DW3x3 -> 1x1 -> DW3x3 -> 1x1 (each + SiLU, rounded to bf16) -> 1x1 + bias, fp32 logits;
    with y: their sigmoid into the class rows of y (+ `y.best`), the class half of detect_decode."""
    _count("detect_cls_fused")
    h = dwconv2d(x, d1[0], d1[1], 3, True)
    h = conv2d(h, p1[0], p1[1], 1, 1, True)
    h = dwconv2d(h, d2[0], d2[1], 3, True)
    h = conv2d(h, p2[0], p2[1], 1, 1, True)
    lg = conv2d(h, w3[0], w3[1], 1, 1, False, out=out if (raw or y is None) else None, out_dtype=torch.float32)
    if y is not None:
        B, H, W, _ = lg.shape
        n = H * W
        y[:, 4:, a_off: a_off + n] = lg[..., :nc].reshape(B, n, nc).sigmoid().transpose(1, 2)
        best = getattr(y, "best", None)
        if best is not None:
            best[0][:, a_off: a_off + n], j = y[:, 4:, a_off: a_off + n].max(1)
            best[1][:, a_off: a_off + n] = j.int()
    return lg if (raw or y is None) else None


def detect_box_tail_supported(dtype, cin, reg_max, nc):
    return dtype == torch.bfloat16 and cin == 64 and reg_max == 16 and 1 <= nc <= 96


def detect_box_tail(x, w_packed, bias, y, stride, a_off, reg_max, raw=False):
    """include/ymk.h `ymk_detect_box_tail`: 1x1 (+bias, fp32) -> DFL -> dist2bbox -> rows 0..3 of y."""
    _count("detect_box_tail")
    lg = conv2d(x, w_packed, bias, 1, 1, False, out_dtype=torch.float32)
    B, Hl, Wl, _ = lg.shape
    n = Hl * Wl
    dist = (lg.reshape(B, n, 4, reg_max).softmax(-1) * torch.arange(reg_max, dtype=torch.float32)).sum(-1)
    sy, sx = torch.meshgrid(torch.arange(Hl, dtype=torch.float32) + 0.5, torch.arange(Wl, dtype=torch.float32) + 0.5, indexing="ij")
    anc = torch.stack((sx, sy), -1).view(1, n, 2)
    x1y1, x2y2 = anc - dist[..., :2], anc + dist[..., 2:]
    y[:, :4, a_off: a_off + n] = (torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1) * stride).transpose(1, 2)
    return lg if raw else None


def mlp_fused_supported(dtype, C, hidden):
    return dtype == torch.bfloat16 and (C, hidden) in ((64, 128), (128, 256), (256, 512))


def mlp_fused(x, w1, b1, w2, b2, out=None):
    """include/ymk.h `ymk_mlp_fused`: y = x + W2 SiLU(W1 x + b1) + b2, the hidden tile rounded to the activation type."""
    _count("mlp_fused")
    h = conv2d(x, w1, b1, 1, 1, True)
    return conv2d(h, w2, b2, 1, 1, False, out=out, residual=x)


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
    elif thr < 0:      # dense forward over the top-k set
        retained = route_w > 0
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
    usage = route_w.mean(0)
    un = usage / usage.sum().clamp_min(1e-6)
    return route_w, gate_w, sel, csr_off, csr_pair, torch.cat([usage, (E * (un * un).sum()).view(1)])


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


def area_attn_qkv_supported(dtype, C, heads, N, area):
    return dtype in (torch.bfloat16, torch.float16) and C == heads * 32 and C in (64, 128) and N % area == 0 and 16 <= N // area <= 448


def area_attn_qkv(x, w, b, heads, area, out=None, v_out=None):
    """`ymk_area_attn_qkv`: qkv = w x + b rounded to the 16-bit type (what the unfused convolution stores), then area_attn; returns (out, v)."""
    _count("area_attn_qkv")
    B, H, W, C = x.shape
    qkv = (x.float().reshape(-1, C) @ w[:, :C].float().t() + b.float()).to(x.dtype).reshape(B, H, W, 3 * C)
    o = area_attn(qkv, heads, area)
    CALLS["area_attn"] -= 1
    v = qkv[..., 2 * C:].contiguous()
    if out is not None:
        out.copy_(o)
        o = out
    if v_out is not None:
        v_out.copy_(v)
        v = v_out
    return o, v


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
def detect_decode(box_l, cls_l, y, stride, a_off, reg_max, best=None):
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
    if best is not None:
        best[0][:, a_off: a_off + n], j = y[:, 4:, a_off: a_off + n].max(1)
        best[1][:, a_off: a_off + n] = j.int()
    return y


def nms_gather_rows(y, nc, idx, counts, out=None):
    _count("nms_gather_rows")
    B, ch, A = y.shape
    extra, max_det = ch - 4 - nc, idx.shape[1]
    res = torch.zeros((B, max_det, extra), dtype=torch.float32)
    for b in range(B):
        n = int(counts[b])
        res[b, :n] = y[b, 4 + nc:, idx[b, :n].long()].t()
    return res


def nms_batched(y, conf, iou, multi_label, agnostic, max_det, max_nms, max_wh, cw_sigma=None, cw_pool=3000, class_keep=None, pack=None, nc=0, use_best=True):
    _count("nms_batched")
    assert cw_sigma is None, "CW refinement is checked on the GPU against oracle/_ref"
    B = y.shape[0]
    classes = None if class_keep is None else torch.nonzero(class_keep).view(-1).tolist()
    outs, idxs = nms_ref.non_max_suppression(y.numpy(), conf, iou, multi_label, agnostic, max_det, max_nms, max_wh,
                                             return_idxs=True, classes=classes, nc=nc)
    outs = [np.asarray(o, np.float32)[:, :6] for o in outs]
    if pack is not None:                     # the three outputs carved from one buffer (ops.nms_pack_views), zeroed like the kernel does
        from yolo_master_amd import ops as _ops

        pack.zero_()
        dets, counts, idx = _ops.nms_pack_views(pack, B, max_det)
    else:
        dets = torch.zeros((B, max_det, 6), dtype=torch.float32)
        counts = torch.zeros((B,), dtype=torch.int32)
        idx = torch.zeros((B, max_det), dtype=torch.int32)
    for b in range(B):
        n = len(outs[b])
        counts[b] = n
        dets[b, :n] = torch.from_numpy(np.asarray(outs[b], np.float32))
        idx[b, :n] = torch.from_numpy(np.asarray(idxs[b]).astype(np.int32))
    return dets, counts, idx, torch.zeros((1,), dtype=torch.int32)


# ------------------------------------------------------------------------------------------------- config-5 rows
def _put(y, out, dtype):
    if out is None:
        return y.to(dtype).contiguous()
    assert tuple(out.shape) == tuple(y.shape), f"out {tuple(out.shape)} vs result {tuple(y.shape)}"
    out.copy_(y.to(out.dtype))
    return out


def _act(y, act):
    if act in (False, None):
        return y
    if act in (True, "silu"):
        return F.silu(y)
    if act == "sigmoid":
        return torch.sigmoid(y)
    if act == "gelu":
        return F.gelu(y)
    raise ValueError(act)


def conv2d_act(x, w_packed, bias, k, stride, act, out=None, residual=None, out_dtype=None):
    _count("conv2d_act")
    if act in (False, True, "silu"):
        return conv2d(x, w_packed, bias, k, stride, bool(act), out=out, residual=residual, out_dtype=out_dtype)
    w = _unpack_conv(w_packed, k, x.shape[-1])
    y = _act(F.conv2d(_nchw(x), w, bias, stride, k // 2), act)
    return _finish(y, False, residual, out, out_dtype or x.dtype)


def group_norm(x, groups, weight, bias, eps, act=False, out=None, out_dtype=None, affine_rows=None, residual=None):
    _count("group_norm")
    B, H, W, C = x.shape
    assert C % groups == 0
    y = F.group_norm(_nchw(x), groups, None, None, eps)
    if weight is not None:
        assert weight.dtype == torch.float32
        if affine_rows is not None:
            assert affine_rows.dtype == torch.int32 and tuple(affine_rows.shape) == (B,)
            w, b = weight[affine_rows.long()], bias[affine_rows.long()]
            y = y * w.view(B, C, 1, 1) + b.view(B, C, 1, 1)
        else:
            y = y * weight.view(1, C, 1, 1) + bias.view(1, C, 1, 1)
    y = _act(y, act).permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual.float()
    return _put(y, out, out_dtype or x.dtype)


def layer_norm(x, weight, bias, eps, out=None):
    _count("layer_norm")
    return _put(F.layer_norm(x.float(), (x.shape[-1],), weight, bias, eps), out, x.dtype)


def eltwise_mul(a, b, out=None, act_a=None):
    _count("eltwise_mul")
    return _put(_act(a.float(), act_a) * b.float(), out, a.dtype)


def clamp_add(a, b, limit, out=None):
    _count("clamp_add")
    return _put(a.float().clamp(-float(limit), float(limit)) + b.float(), out, a.dtype)


def lerp(a, b, alpha, out=None):
    _count("lerp")
    return _put((1 - alpha) * a.float() + alpha * b.float(), out, a.dtype)


def fma_gate(x, a, b, scale, out=None):
    _count("fma_gate")
    assert b.shape == x.shape or tuple(b.shape) == (x.shape[0], 1, 1, x.shape[3])
    return _put(x.float() + float(scale) * a.float() * b.float(), out, x.dtype)


def batch_scale(w, logit, lo, hi):
    _count("batch_scale")
    c = torch.sigmoid(logit[:, 0, 0, 0]).mean()
    c = torch.nan_to_num(c, nan=1.0, posinf=1.0, neginf=1.0).clamp(lo, hi) if torch.isfinite(c) else torch.tensor(1.0)
    w.mul_(c)
    return w


def channel_gate(x, gate, out=None):
    _count("channel_gate")
    assert gate.dtype == torch.float32 and tuple(gate.shape) == (x.shape[0], 1, 1, x.shape[3]) and gate.is_contiguous()   # as ops.channel_gate
    return _put(x.float() * gate, out, x.dtype)


def weighted_sum(weights, parts, out=None):
    _count("weighted_sum")
    assert weights.dtype == torch.float32 and weights.shape[-1] >= len(parts)
    assert tuple(weights.shape[:3]) in (tuple(parts[0].shape[:3]), (parts[0].shape[0], 1, 1))
    y = 0
    for e, p in enumerate(parts):
        y = y + weights[..., e:e + 1] * p.float()
    return _put(y, out, parts[0].dtype)


def mean_upsampled(parts, out=None):
    _count("mean_upsampled")
    H, W = parts[0].shape[1:3]
    y = 0
    for p in parts:
        y = y + (p.float() if tuple(p.shape[1:3]) == (H, W) else
                 F.interpolate(_nchw(p), size=(H, W), mode="nearest").permute(0, 2, 3, 1))
    return _put(y / len(parts), out, parts[0].dtype)


def adaptive_avg_pool(x, Ho, Wo, out=None, out_dtype=None):
    _count("adaptive_avg_pool")
    return _put(F.adaptive_avg_pool2d(_nchw(x), (Ho, Wo)).permute(0, 2, 3, 1), out, out_dtype or x.dtype)


def avg_pool(x, k, out=None, out_dtype=None):
    _count("avg_pool")
    return _put(F.avg_pool2d(_nchw(x), k, k).permute(0, 2, 3, 1), out, out_dtype or x.dtype)


def channel_stats(x, want_std=False):
    _count("channel_stats")
    xf = x.float()
    mean = xf.mean((1, 2), keepdim=True)
    if not want_std:
        return mean
    std = xf.std((1, 2), unbiased=False, keepdim=True) if x.shape[1] * x.shape[2] > 1 else torch.zeros_like(mean)
    return torch.cat([mean, std], -1)


def _heads(t, heads, hd):
    B, H, W, C = t.shape
    assert C == heads * hd
    return t.float().reshape(B, H * W, heads, hd).permute(0, 2, 1, 3)   # [B, heads, N, hd]


def attention(q, k, v, heads, hd, scale, out=None):
    _count("attention")
    assert hd % 8 == 0, "kernel contract: head dims are padded to a multiple of 8 by the host"
    B, H, W, _ = q.shape
    p = torch.softmax(_heads(q, heads, hd) @ _heads(k, heads, hd).transpose(-1, -2) * scale, -1)
    o = (p @ _heads(v, heads, hd)).permute(0, 2, 1, 3).reshape(B, H, W, heads * hd)
    return _put(o, out, q.dtype)


def window_attention(q, k, v, heads, hd, scale, win, shift=0, pad_q=None, pad_k=None, pad_v=None, out=None):
    _count("window_attention")
    assert hd % 8 == 0
    B, H, W, C = q.shape
    ph, pw = (win - H % win) % win, (win - W % win) % win
    Hp, Wp = H + ph, W + pw

    def prep(t, padv):
        t = F.pad(t.float(), (0, 0, 0, pw, 0, ph))
        if padv is not None and (ph or pw):
            m = torch.ones(Hp, Wp, dtype=torch.bool)
            m[:H, :W] = False
            t[:, m] = padv.float()
        if shift:
            t = torch.roll(t, (-shift, -shift), (1, 2))
        t = t.reshape(B, Hp // win, win, Wp // win, win, heads, hd).permute(0, 1, 3, 5, 2, 4, 6)
        return t.reshape(B, Hp // win, Wp // win, heads, win * win, hd)

    qw, kw, vw = prep(q, pad_q), prep(k, pad_k), prep(v, pad_v)
    o = torch.softmax(qw @ kw.transpose(-1, -2) * scale, -1) @ vw
    o = o.reshape(B, Hp // win, Wp // win, heads, win, win, hd).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, Hp, Wp, C)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return _put(o[:, :H, :W], out, q.dtype)


def linear_attention(q, k, v, rf, heads, hd, out=None):
    _count("linear_attention")
    B, H, W, _ = q.shape
    assert rf.dtype == torch.float32 and rf.shape[1] == hd
    nb = rf.shape[0]
    qh, kh, vh = _heads(q, heads, hd), _heads(k, heads, hd), _heads(v, heads, hd)
    qf = (F.relu(qh @ rf.t() * nb ** -0.5) + 1e-6).clamp(max=1e4)
    kf = (F.relu(kh @ rf.t() * nb ** -0.5) + 1e-6).clamp(max=1e4)
    numer = (qf @ (kf.transpose(-1, -2) @ vh)).clamp(-1e4, 1e4)
    denom = (qf @ kf.sum(2).unsqueeze(-1)).clamp(min=1e-6)
    return _put((numer / denom).permute(0, 2, 1, 3).reshape(B, H, W, heads * hd), out, q.dtype)


def deform_attention(v, off_logits, aw_logits, heads, hd, n_points, align_corners, out=None):
    _count("deform_attention")
    B, H, W, C = v.shape
    N = H * W
    assert C == heads * hd and off_logits.shape[-1] == heads * n_points * 2 and aw_logits.shape[-1] == heads * n_points
    off = off_logits.float().reshape(B, N, heads, n_points, 2).tanh()
    aw = torch.softmax(aw_logits.float().reshape(B, N, heads, n_points), -1)
    idx = torch.arange(N)
    row = (idx // W).float() / max(H - 1, 1) * 2 - 1
    col = (idx % W).float() / max(W - 1, 1) * 2 - 1
    ref = torch.stack([col, row], -1)[None, :, None, None, :]
    locs = (ref + off * 0.25).clamp(-1.0, 1.0).permute(0, 2, 1, 3, 4).reshape(B * heads, N, n_points, 2)
    v4 = v.float().reshape(B, N, heads, hd).permute(0, 2, 3, 1).reshape(B * heads, hd, H, W)
    smp = F.grid_sample(v4, locs, mode="bilinear", padding_mode="zeros", align_corners=align_corners)   # [B*h, hd, N, np]
    smp = smp.reshape(B, heads, hd, N, n_points).permute(0, 3, 1, 4, 2)                                    # [B, N, h, np, hd]
    o = (aw.unsqueeze(-1) * smp).sum(3).reshape(B, H, W, C)
    return _put(o, out, v.dtype)


def token_softmax(logits, n, inv_temp, top_k=0, out=None, bias=None, shape=None):
    _count("token_softmax")
    if logits is None:
        B, H, W = shape
        lg = torch.zeros((B, H, W, n), dtype=torch.float32)
    else:
        assert logits.dtype == torch.float32
        lg = logits[..., :n]
    if bias is not None:
        lg = lg + bias.view(bias.shape[0], 1, 1, n)
    w = torch.softmax(lg * inv_temp, -1)
    if 0 < top_k < n:
        vals, idx = w.topk(top_k, -1)
        w = torch.zeros_like(w).scatter_(-1, idx, vals / vals.sum(-1, keepdim=True).clamp_min(1e-6))
        sel = torch.zeros_like(w, dtype=torch.bool).scatter_(-1, idx, True)
    else:
        sel = torch.ones_like(w, dtype=torch.bool)
    active = sel.reshape(w.shape[0], -1, n).any(1).to(torch.int32)
    return _put(w, out, torch.float32), active


def scene_bias(x, w1, b1, w2, b2, base=None):
    """include/ymk_mixture.h `ymk_scene_bias`: scene statistics (mot/router.py:166-192) + projector."""
    _count("scene_bias")
    from oracle import mot_ref

    stats = mot_ref.compute_scene_stats(x.permute(0, 3, 1, 2))
    bias = F.linear(F.silu(F.linear(stats, w1, b1)), w2, b2)
    return stats, (bias if base is None else bias + base)


def moa_sparse_gate(weights, n, threshold):
    _count("moa_sparse_gate")
    w = weights[..., :n]
    active = w.amax(dim=(0, 1, 2)) > threshold
    if not bool(active.any()):
        active = torch.zeros_like(active)
        active[w.mean(dim=(0, 1, 2)).argmax()] = True
    bw = w * active.view(1, 1, 1, -1)
    bw = bw / bw.sum(dim=3, keepdim=True).clamp_min(torch.finfo(torch.float32).eps)
    blend = torch.zeros_like(w)
    idx = [g for g in range(n) if bool(active[g])]
    blend[..., :len(idx)] = bw[..., idx]
    mass = float(w[..., ~active].sum(dim=3).mean()) if not bool(active.all()) else 0.0
    return [bool(a) for a in active.tolist()], blend.contiguous(), mass


def gated_route_decide(g_logits, loc_logits, alpha, inv_temp, top_k, cplx_logit, clamp=1):
    _count("gated_route_decide")
    B, E = g_logits.shape[0], g_logits.shape[-1]
    a = 1.0 / (1.0 + float(np.exp(-alpha)))
    logits = a * g_logits.reshape(B, E) + (1 - a) * loc_logits.reshape(B, E)
    if clamp == 1:
        logits = logits.clamp(-30.0, 30.0)
    logits = logits * inv_temp
    if clamp == 2:
        logits = logits.clamp(-30.0, 30.0)
    probs = torch.softmax(logits, 1)
    tw, ti = torch.topk(probs, top_k, 1)
    tw = tw / (tw.sum(1, keepdim=True) + 1e-6)
    c = torch.sigmoid(cplx_logit.reshape(B)).mean()
    c = torch.tensor(1.0) if not bool(torch.isfinite(c)) else c.clamp(0.3, 1.5)
    if top_k > 1:
        keep = torch.round(c * top_k).clamp(1, top_k)
        tw = tw * (torch.arange(1, top_k + 1).view(1, -1) <= keep).float()
        tw = tw / tw.sum(1, keepdim=True).clamp_min(1e-6)
    ti = ti.to(torch.int32)
    return tw.reshape(B, 1, 1, top_k), ti, probs, ti.t().contiguous().reshape(-1)


def pooled_softmax_route(logits, E, inv_temp, top_k, threshold):
    """include/ymk_mixture.h `ymk_pooled_softmax_route`: per-pixel softmax of the clamped logits, mean over the pixels, top-k, renormalise,
    weights <= threshold zeroed."""
    _count("pooled_softmax_route")
    B = logits.shape[0]
    sm = torch.softmax(logits[..., :E].float().clamp(-30.0, 30.0) * inv_temp, dim=-1)
    pooled = sm.reshape(B, -1, E).mean(1)
    tw, ti = torch.topk(pooled, top_k, 1)
    tw = tw / tw.sum(1, keepdim=True).clamp_min(1e-6)
    tw = torch.where(tw > threshold, tw, torch.zeros_like(tw))
    ti = ti.to(torch.int32)
    return tw.reshape(B, 1, 1, top_k), ti, pooled, ti.t().contiguous().reshape(-1)


def expert_conv(x, w_packed, k, idx, out=None):
    _count("expert_conv")
    B, H, W, Cin = x.shape
    E, Cout, Kp = w_packed.shape
    K = idx.shape[1]
    assert idx.dtype == torch.int32 and w_packed.dtype == x.dtype
    ys = []
    for j in range(K):            # slot-major: out[j*B + b]
        for b in range(B):
            w = _unpack_conv(w_packed[int(idx[b, j])], k, Cin)
            ys.append(F.conv2d(_nchw(x[b:b + 1]), w, None, 1, k // 2))
    return _put(torch.cat(ys).permute(0, 2, 3, 1), out, x.dtype)


def expert_dw3(x, w, dil, idx, out=None):
    """include/ymk_mixture.h `ymk_expert_dw3`."""
    _count("expert_dw3")
    B, H, W, C = x.shape
    K = idx.shape[1]
    xn = x.float().permute(0, 3, 1, 2)
    ys = []
    for j in range(K):
        for b in range(B):
            e = int(idx[b, j])
            d = int(dil[e])
            wk = w[e].float().t().reshape(C, 1, 3, 3)
            ys.append(F.conv2d(xn[b:b + 1], wk, None, 1, d, d, C))
    return _put(torch.cat(ys).permute(0, 2, 3, 1), out, x.dtype)


def channel_shuffle_cat(parts, groups, out=None):
    _count("channel_shuffle_cat")
    cat = torch.cat(parts, -1)
    B, H, W, C = cat.shape
    y = cat.reshape(B, H, W, groups, C // groups).transpose(3, 4).reshape(B, H, W, C)
    return _put(y, out, cat.dtype)


def pixel_shuffle2(t, out=None):
    _count("pixel_shuffle2")
    B, H, W, C4 = t.shape
    c = C4 // 4
    y = t.reshape(B, H, W, 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, c)
    return _put(y, out, t.dtype)


def tokens_to_rows(x, y, a_off, row_off=0):
    _count("tokens_to_rows")
    B, H, W, c = x.shape
    y[:, row_off: row_off + c, a_off: a_off + H * W] = x.float().reshape(B, H * W, c).transpose(1, 2)
    return y


EMULATED = ["conv2d", "conv1x1_cat2", "conv2d_stem", "dwconv2d", "mlp_fused_supported", "mlp_fused", "stem_pair_supported", "stem_pair", "c3k2_fused_supported", "c3k2_fused", "detect_cls_fused_supported", "detect_cls_fused", "detect_box_tail_supported", "detect_box_tail", "esmoe_route", "esmoe_dw",
            "esmoe_pw", "area_attn", "area_attn_qkv_supported", "area_attn_qkv", "upsample2x", "copy_channels", "scale_residual", "nhwc_to_nchw_f32",
            "detect_decode", "nms_batched", "nms_gather_rows",
            "conv2d_act", "group_norm", "layer_norm", "eltwise_mul", "lerp", "fma_gate", "channel_gate", "batch_scale", "weighted_sum",
            "mean_upsampled", "adaptive_avg_pool", "avg_pool", "channel_stats", "attention", "window_attention",
            "linear_attention", "deform_attention", "token_softmax", "scene_bias", "moa_sparse_gate", "gated_route_decide", "pooled_softmax_route", "clamp_add", "expert_conv", "expert_dw3", "channel_shuffle_cat",
            "pixel_shuffle2", "tokens_to_rows"]
