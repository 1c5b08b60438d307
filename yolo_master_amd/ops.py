"""Tensor-level wrappers over the libymk C-ABI (include/ymk.h).

Activations are NHWC torch tensors ``[B, H, W, C]`` on the GPU whose channel dim is dense
(stride 1) and whose pixel stride ``stride(2)`` may exceed ``C`` (a channel slice of a wider
concat buffer).  Every wrapper launches on ``torch.cuda.current_stream()`` and never
synchronises.  torch is used for memory and streams only — all arithmetic is in libymk.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvDesc, check, lib

DT = {torch.float32: _lib.YMK_F32, torch.bfloat16: _lib.YMK_BF16}


class KernelTimer:
    """Optional per-call HIP-event timing of the op families (bench.py's roofline leg, tools/gpu_diag.py).
    Events are recorded on the stream the kernels are launched on (torch's current stream).  Inactive unless
    `start()` was called; never active inside a captured graph."""

    def __init__(self):
        self.on = False
        self.records = []  # (family, start_event, end_event, algorithmic_bytes, flops)
        self.shapes = []   # free-form shape string per record (tools/gpu_diag.py calls)

    def start(self):
        self.on, self.records, self.shapes = True, [], []

    def stop(self):
        self.on = False

    def begin(self):
        if not self.on:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def end(self, e0, family, nbytes, flops, shape=""):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((family, e0, e1, int(nbytes), int(flops)))
        self.shapes.append(shape)


TIMER = KernelTimer()
CONV_FAMILY = {0: "conv_igemm", 1: "conv1x1_ws", 2: "conv3x3_tile"}  # ymk_conv2d_last_variant()


def _tile(cout: int) -> str:
    """cout x pixel tile and wave grid of conv_igemm_kernel (csrc/conv.hip launch_conv / launch_conv_dual)."""
    return "128, 128, 2, 2" if cout > 64 else "64, 256, 1, 4" if cout > 32 else "32, 256, 1, 4" if cout > 16 else "16, 256, 1, 4"


def conv_kernel_name(variant: int, dtype, cin: int, cout: int, k: int, kpad: int, residual: bool, dual: bool = False) -> str:
    """Demangled name of the kernel instantiation a ymk_conv2d / ymk_conv1x1_cat2 call ran (as rocprofv3 prints it,
    with bf16_t = unsigned short): the timer tags conv calls with it so that bench.py's roofline object and the
    committed rocprof / PMC summaries refer to the same kernel."""
    t = "unsigned short" if dtype == torch.bfloat16 else "float"
    es = 2 if dtype == torch.bfloat16 else 4
    if dual:
        return f"conv_igemm_kernel<{t}, {_tile(cout)}, 1, true>"
    if variant == 1:
        return f"conv1x1_ws_kernel<{t}, {kpad * es // 128}>"
    if variant == 2:
        prefetch = residual and not (int(os.environ.get("YMK_DISABLE", "0"), 0) & 16)
        return f"conv3x3_tile_kernel<{t}, {cin}, {32 if cout <= 32 else 64}, {'true' if prefetch else 'false'}>"
    return f"conv_igemm_kernel<{t}, {_tile(cout)}, {k}, false>"


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def require_gpu(t: torch.Tensor, what: str = "yolo_master_amd ops") -> None:
    """The one device guard of the product (modules, graph walk and op wrappers all call it): no CPU fallback."""
    if not t.is_cuda:
        raise RuntimeError(f"{what} run on MI355X (HIP) only; got a CPU tensor. There is no CPU fallback.")


def _need_gpu(t: torch.Tensor) -> None:
    require_gpu(t)


def _nhwc(t: torch.Tensor):
    """Validate an NHWC view; return (B, H, W, C, ld)."""
    _need_gpu(t)
    B, H, W, Cc = t.shape
    if t.stride(3) != 1 and Cc > 1:
        raise ValueError("NHWC view must be channel-dense")
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else max(Cc, t.stride(0) // max(H * W, 1)))
    if W > 1 and H > 1 and t.stride(1) != W * ld:
        raise ValueError("NHWC view rows must be contiguous in pixels")
    if B > 1 and t.stride(0) != H * W * ld:
        raise ValueError("NHWC view images must be contiguous in pixels")
    return B, H, W, Cc, ld


def new_act(B: int, H: int, W: int, Cc: int, dtype: torch.dtype, device) -> torch.Tensor:
    return torch.empty((B, H, W, Cc), dtype=dtype, device=device)


# ----------------------------------------------------------------------------- packing
def kpad(k: int) -> int:
    return (k + 63) // 64 * 64


def pack_conv_weight(w: torch.Tensor, dtype: torch.dtype, cout_perm: torch.Tensor | None = None) -> torch.Tensor:
    """[Cout, Cin, kh, kw] fp32 -> [Cout, Kpad] with K ordered (ky, kx, cin), zero padded."""
    co, ci, kh, kw = w.shape
    m = w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
    if cout_perm is not None:
        m = m[cout_perm]
    out = torch.zeros((co, kpad(kh * kw * ci)), dtype=torch.float32, device=w.device)
    out[:, : kh * kw * ci] = m
    return out.to(dtype).contiguous()


def pack_dw_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[C, 1, k, k] -> [k*k, C]."""
    c, _, kh, kw = w.shape
    return w.reshape(c, kh * kw).t().contiguous().to(dtype)


def fold_bn(w: torch.Tensor, bn_w, bn_b, bn_mean, bn_var, eps: float, conv_bias=None):
    """BN fold exactly as fuse_conv_and_bn (ultralytics/utils/torch_utils.py:315-349)."""
    co = w.shape[0]
    w_bn = torch.diag(bn_w.div(torch.sqrt(eps + bn_var)))
    wf = torch.mm(w_bn, w.reshape(co, -1)).view(w.shape)
    b_conv = torch.zeros(co, device=w.device, dtype=w.dtype) if conv_bias is None else conv_bias
    b_bn = bn_b - bn_w.mul(bn_mean).div(torch.sqrt(bn_var + eps))
    bf = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn
    return wf, bf


# ----------------------------------------------------------------------------- conv
def conv2d(x, w_packed, bias, k: int, stride: int, act: bool, out=None, residual=None, out_dtype=None):
    B, H, W, Cin, ldx = _nhwc(x)
    Cout, Kp = w_packed.shape
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    odt = out_dtype or x.dtype
    if out is None:
        out = new_act(B, Ho, Wo, Cout, odt, x.device)
    Bo, Ho2, Wo2, Co2, ldy = _nhwc(out)
    assert (Bo, Ho2, Wo2, Co2) == (B, Ho, Wo, Cout), f"conv out shape {tuple(out.shape)} != {(B, Ho, Wo, Cout)}"
    ldr = 0
    if residual is not None:
        rb = _nhwc(residual)
        assert rb[:4] == (B, Ho, Wo, Cout)
        ldr = rb[4]
    d = ConvDesc(DT[x.dtype], DT[out.dtype], B, H, W, Cin, Cout, k, stride, ldx, ldy, ldr, Kp,
                 _lib.ACT_SILU if act else _lib.ACT_NONE)
    e0 = TIMER.begin()
    check(lib.ymk_conv2d(C.byref(d), _p(x), _p(w_packed), _p(bias), _p(residual), _p(out), _stream()), "conv2d")
    if e0 is not None:
        es = x.element_size()
        nbytes = (B * H * W * Cin + Cout * k * k * Cin + (B * Ho * Wo * Cout if residual is not None else 0)) * es \
            + B * Ho * Wo * Cout * out.element_size()
        name = conv_kernel_name(lib.ymk_conv2d_last_variant(), x.dtype, Cin, Cout, k, Kp, residual is not None)
        TIMER.end(e0, name, nbytes, 2 * B * Ho * Wo * Cout * k * k * Cin,
                  f"{Cin}->{Cout} k{k} s{stride} @{Ho}x{Wo}{' +res' if residual is not None else ''}")
    return out


def conv1x1_cat2(x1, up1: bool, x2, w_packed, bias, act: bool, out=None):
    """1x1 conv over cat([upsample2x(x1) if up1 else x1, x2], channel) without building the concatenation."""
    B1, H1, W1, C1, ld1 = _nhwc(x1)
    B, H, W, C2, ld2 = _nhwc(x2)
    assert B1 == B and (H1 * (2 if up1 else 1), W1 * (2 if up1 else 1)) == (H, W) and x1.dtype == x2.dtype
    Cout, Kp = w_packed.shape
    if out is None:
        out = new_act(B, H, W, Cout, x2.dtype, x2.device)
    ldy = _nhwc(out)[4]
    d = ConvDesc(DT[x2.dtype], DT[out.dtype], B, H, W, C1 + C2, Cout, 1, 1, ld1, ldy, 0, Kp,
                 _lib.ACT_SILU if act else _lib.ACT_NONE)
    e0 = TIMER.begin()
    check(lib.ymk_conv1x1_cat2(C.byref(d), _p(x1), C1, ld1, int(up1), _p(x2), ld2, _p(w_packed), _p(bias), _p(out), _stream()),
          "conv1x1_cat2")
    es = x2.element_size()
    TIMER.end(e0, conv_kernel_name(0, x2.dtype, C1 + C2, Cout, 1, Kp, False, dual=True),
              (B1 * H1 * W1 * C1 + B * H * W * C2 + Cout * (C1 + C2) + B * H * W * Cout) * es,
              2 * B * H * W * Cout * (C1 + C2), f"{C1}+{C2}->{Cout} @{H}x{W}{' up' if up1 else ''}")
    return out


def conv2d_stem(x_nchw, w, bias, k: int, stride: int, act: bool, dtype: torch.dtype, out=None, wt=None):
    _need_gpu(x_nchw)
    x_nchw = x_nchw.contiguous().float()
    B, Cin, H, W = x_nchw.shape
    Cout = w.shape[0]
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = new_act(B, Ho, Wo, Cout, dtype, x_nchw.device)
    ldy = _nhwc(out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_conv2d_stem_nchw(_p(x_nchw), _p(w), _p(wt), _p(bias), _p(out), DT[out.dtype], B, Cin, H, W, Cout, k, stride,
                                   ldy, _lib.ACT_SILU if act else _lib.ACT_NONE, _stream()), "conv2d_stem_nchw")
    TIMER.end(e0, "stem", B * Cin * H * W * 4 + B * Ho * Wo * Cout * out.element_size(), 2 * B * Ho * Wo * Cout * k * k * Cin)
    return out


def dwconv2d(x, w_packed, bias, k: int, act: bool, out=None, residual=None):
    B, H, W, Cc, ldx = _nhwc(x)
    if out is None:
        out = new_act(B, H, W, Cc, x.dtype, x.device)
    ldy = _nhwc(out)[4]
    ldr = _nhwc(residual)[4] if residual is not None else 0
    e0 = TIMER.begin()
    check(lib.ymk_dwconv2d(DT[x.dtype], _p(x), _p(w_packed), _p(bias), _p(residual), _p(out), B, H, W, Cc, k, ldx, ldy,
                           ldr, _lib.ACT_SILU if act else _lib.ACT_NONE, _stream()), "dwconv2d")
    TIMER.end(e0, "dwconv", B * H * W * Cc * x.element_size() * (3 if residual is not None else 2), 2 * B * H * W * Cc * k * k, f"C{Cc} k{k} @{H}x{W}")
    return out


# ----------------------------------------------------------------------------- ES-MoE
def esmoe_route(x, w1, b1, w2, b2, top_k: int, thr: float, flags: torch.Tensor):
    B, H, W, Cc, ldx = _nhwc(x)
    hidden, E = w1.shape[0], w2.shape[0]
    dev = x.device
    route_w = torch.empty((B, E), dtype=torch.float32, device=dev)
    gate_w = torch.empty((B, E), dtype=torch.float32, device=dev)
    sel = torch.empty((B, top_k), dtype=torch.int32, device=dev)
    csr_off = torch.empty((E + 1,), dtype=torch.int32, device=dev)
    csr_pair = torch.empty((B * top_k,), dtype=torch.int32, device=dev)
    nbytes = lib.ymk_esmoe_route_workspace_bytes(B, Cc, H, W)
    ws = torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=dev)
    e0 = TIMER.begin()
    check(lib.ymk_esmoe_route(DT[x.dtype], _p(x), B, H, W, Cc, ldx, _p(w1), _p(b1), _p(w2), _p(b2), hidden, E, top_k,
                              float(thr), _p(route_w), _p(gate_w), _p(sel), _p(csr_off), _p(csr_pair), _p(flags),
                              _p(ws), nbytes, _stream()), "esmoe_route")
    TIMER.end(e0, "moe_route", B * H * W * Cc * x.element_size(), B * H * W * Cc, f"C{Cc} @{H}x{W}")
    return route_w, gate_w, sel, csr_off, csr_pair


def esmoe_dw(x, dw_w, dw_off, ksizes, kmax: int, top_k: int, sel, csr_off, csr_pair):
    B, H, W, Cc, ldx = _nhwc(x)
    E = ksizes.numel()
    out = torch.empty((B * top_k, H, W, Cc), dtype=x.dtype, device=x.device)
    e0 = TIMER.begin()
    check(lib.ymk_esmoe_dw(DT[x.dtype], _p(x), B, H, W, Cc, ldx, _p(dw_w), _p(dw_off), _p(ksizes), E, top_k, kmax, _p(sel),
                           _p(csr_off), _p(csr_pair), _p(out), _stream()), "esmoe_dw")
    if e0 is not None:  # algorithmic traffic: every image read once, one plane written per retained (image, expert) pair
        e1 = TIMER.begin()
        cnt = (csr_off[1:] - csr_off[:-1]).cpu()
        npairs, k2 = int(cnt.sum()), int((cnt * ksizes.cpu().int() ** 2).sum())
        TIMER.records.append(("moe_dw", e0, e1, (B + npairs) * H * W * Cc * x.element_size(), 2 * k2 * H * W * Cc))
        TIMER.shapes.append(f"C{Cc} @{H}x{W} pairs {npairs}")
    return out


def esmoe_pw(dw_out, B: int, H: int, W: int, pw_w, pw_b, nscale, nshift, top_k: int, sel, gate_w, out=None):
    Cc = dw_out.shape[-1]
    E, Cout, Kp = pw_w.shape
    if out is None:
        out = new_act(B, H, W, Cout, dw_out.dtype, dw_out.device)
    ldy = _nhwc(out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_esmoe_pw(DT[dw_out.dtype], _p(dw_out), B, H, W, Cc, Cout, Kp, _p(pw_w), _p(pw_b), _p(nscale),
                           _p(nshift), E, top_k, _p(sel), _p(gate_w), _p(out), ldy, _stream()), "esmoe_pw")
    if e0 is not None:  # retained (image, expert) pairs: sel is -1 for dropped slots
        e1 = TIMER.begin()
        npairs = int((sel >= 0).sum())
        es = dw_out.element_size()
        TIMER.records.append(("moe_pw", e0, e1, (npairs * H * W * Cc + E * Cout * Cc + B * H * W * Cout) * es,
                              2 * npairs * H * W * Cc * Cout))
        TIMER.shapes.append(f"{Cc}->{Cout} @{H}x{W} pairs {npairs}")
    return out


def dwpw_supported(dtype, C: int, kmax: int) -> bool:
    return bool(lib.ymk_dwpw_supported(DT[dtype], C, kmax))


def esmoe_experts_fused(x, dw_w, dw_off, ksizes, kmax: int, pw_w, pw_b, nscale, nshift, top_k: int, sel, gate_w, out=None):
    B, H, W, Cc, ldx = _nhwc(x)
    E, Cout, Kp = pw_w.shape
    if out is None:
        out = new_act(B, H, W, Cout, x.dtype, x.device)
    ldy = _nhwc(out)[4]
    check(lib.ymk_esmoe_experts_fused(DT[x.dtype], _p(x), B, H, W, Cc, ldx, _p(dw_w), _p(dw_off), _p(ksizes), kmax, Cout, Kp,
                                      _p(pw_w), _p(pw_b), _p(nscale), _p(nshift), E, top_k, _p(sel), _p(gate_w), _p(out), ldy,
                                      _stream()), "esmoe_experts_fused")
    return out


def dwconv_pwconv(x, dw_w, dw_b, k: int, dw_act: bool, pw_w, pw_b, pw_act: bool, out=None):
    B, H, W, Cc, ldx = _nhwc(x)
    Cout, Kp = pw_w.shape
    if out is None:
        out = new_act(B, H, W, Cout, x.dtype, x.device)
    ldy = _nhwc(out)[4]
    check(lib.ymk_dwconv_pwconv(DT[x.dtype], _p(x), B, H, W, Cc, ldx, _p(dw_w), _p(dw_b), k, int(dw_act), Cout, Kp, _p(pw_w),
                                _p(pw_b), int(pw_act), _p(out), ldy, _stream()), "dwconv_pwconv")
    return out


# ----------------------------------------------------------------------------- attention
def area_attn(qkv, heads: int, area: int, out=None):
    B, H, W, C3, ldq = _nhwc(qkv)
    Cq = heads * 32
    assert C3 == 3 * Cq
    if out is None:
        out = new_act(B, H, W, Cq, qkv.dtype, qkv.device)
    ldo = _nhwc(out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_area_attn(DT[qkv.dtype], _p(qkv), ldq, _p(out), ldo, B, H * W, heads, area, _stream()), "area_attn")
    TIMER.end(e0, "area_attn", B * H * W * 4 * Cq * qkv.element_size(), 4 * B * H * W * (H * W // area) * Cq,
              f"heads {heads} area {area} @{H}x{W}")
    return out


# ----------------------------------------------------------------------------- layout
def upsample2x(x, out=None):
    B, H, W, Cc, ldx = _nhwc(x)
    if out is None:
        out = new_act(B, 2 * H, 2 * W, Cc, x.dtype, x.device)
    ldy = _nhwc(out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_upsample2x(DT[x.dtype], _p(x), _p(out), B, H, W, Cc, ldx, ldy, _stream()), "upsample2x")
    TIMER.end(e0, "layout", 5 * B * H * W * Cc * x.element_size(), 0)
    return out


def copy_channels(x, out):
    B, H, W, Cc, ldx = _nhwc(x)
    ldy = _nhwc(out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_copy_channels(DT[x.dtype], _p(x), _p(out), B * H * W, Cc, ldx, ldy, _stream()), "copy_channels")
    TIMER.end(e0, "layout", 2 * B * H * W * Cc * x.element_size(), 0)
    return out


def scale_residual(y, gamma, residual, out=None):
    """out = residual + gamma[c] * y (A2C2f gamma-residual, block.py:1877-1879)."""
    B, H, W, Cc, ldy = _nhwc(y)
    ldr = _nhwc(residual)[4]
    if out is None:
        out = new_act(B, H, W, Cc, y.dtype, y.device)
    ldo = _nhwc(out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_scale_residual(DT[y.dtype], _p(y), _p(gamma), _p(residual), _p(out), B * H * W, Cc, ldy, ldr, ldo, _stream()),
          "scale_residual")
    TIMER.end(e0, "layout", 3 * B * H * W * Cc * y.element_size(), 2 * B * H * W * Cc)
    return out


def nhwc_to_nchw_f32(x):
    B, H, W, Cc, ldx = _nhwc(x)
    y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    e0 = TIMER.begin()
    check(lib.ymk_nhwc_to_nchw_f32(DT[x.dtype], _p(x), _p(y), B, H * W, Cc, ldx, _stream()), "nhwc_to_nchw_f32")
    TIMER.end(e0, "layout", B * H * W * Cc * (x.element_size() + 4), 0)
    return y


# ----------------------------------------------------------------------------- detect / nms
def detect_decode(box_l, cls_l, y, stride: float, a_off: int, reg_max: int):
    B, Hl, Wl, _, _ = _nhwc(box_l)
    nc = cls_l.shape[-1]
    assert box_l.is_contiguous() and cls_l.is_contiguous() and box_l.dtype == torch.float32
    e0 = TIMER.begin()
    check(lib.ymk_detect_decode(_p(box_l), _p(cls_l), _p(y), B, Hl, Wl, reg_max, nc, float(stride), a_off, y.shape[2],
                                _stream()), "detect_decode")
    TIMER.end(e0, "detect_decode", B * Hl * Wl * (4 * reg_max + nc + 4 + nc) * 4, B * Hl * Wl * (4 * reg_max * 4 + nc * 4))
    return y


def nms_batched(y, conf: float, iou: float, multi_label: bool, agnostic: bool, max_det: int, max_nms: int,
                max_wh: float, cw_sigma: float | None = None, cw_pool: int = 3000):
    """Returns (dets [B,max_det,6], counts [B] int32, idx [B,max_det] int32, status [1] int32)."""
    _need_gpu(y)
    assert y.dtype == torch.float32 and y.is_contiguous()
    B, ch, A = y.shape
    nc = ch - 4
    dev = y.device
    nbytes = lib.ymk_nms_workspace_bytes(B, nc, A, int(multi_label), max_nms)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    dets = torch.zeros((B, max_det, 6), dtype=torch.float32, device=dev)
    counts = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx = torch.zeros((B, max_det), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    e0 = TIMER.begin()
    check(lib.ymk_nms_batched(_p(y), B, nc, A, float(conf), float(iou), int(multi_label), int(agnostic), max_det, max_nms,
                              float(max_wh), _p(dets), _p(counts), _p(idx), _p(status), _p(ws), nbytes, _stream()),
          "nms_batched")
    if cw_sigma is not None:
        check(lib.ymk_cw_refine(B, nc, A, int(multi_label), max_nms, max_det, float(iou), float(cw_sigma), cw_pool,
                                _p(dets), _p(counts), _p(ws), nbytes, _stream()), "cw_refine")
    TIMER.end(e0, "nms", B * ch * A * 4 + B * max_det * 28, B * ch * A)
    return dets, counts, idx, status
