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

import threading

from . import _lib
from ._lib import ConvDesc, check
from .options import OPTIONS

# Element types.  libymk has ONE 16-bit element type per build (include/ymk.h YMK_H16): bfloat16 in libymk.so, IEEE binary16 in
# libymk_f16.so (the same sources and C-ABI compiled with -DYMK_H16_F16) — the reference's `half=True` mode is fp16.
H16 = (torch.bfloat16, torch.float16)
DT = {torch.float32: _lib.YMK_F32, torch.bfloat16: _lib.YMK_H16, torch.float16: _lib.YMK_H16}
_TLS = threading.local()


class _LibRouter:
    """`lib.ymk_<entry point>`: libymk.so, or libymk_f16.so when the last 16-bit NHWC view validated on this thread (`_nhwc`, which every
    wrapper calls on its tensors before it touches the library) was torch.float16.  fp32-only calls are identical in both builds."""

    def __getattr__(self, name):
        return getattr(_lib.load_f16() if getattr(_TLS, "fmt", None) is torch.float16 else _lib.load(), name)


lib = _LibRouter()


def use_format(dtype: torch.dtype) -> None:
    """Select the library build for the calls that follow on this thread (a model walk calls it once with its compute dtype;
    `_nhwc` keeps it in step with the tensors actually passed)."""
    if dtype in H16:
        _TLS.fmt = dtype


class KernelTimer:
    """Optional per-call HIP-event timing of the op families (bench.py's roofline leg, tools/gpu_diag.py).
    Events are recorded on the stream the kernels are launched on (torch's current stream).  Inactive unless
    `start()` was called; never active inside a captured graph."""

    def __init__(self):
        self.on = False
        self.records = []  # (family, start_event, end_event, algorithmic_bytes, flops)
        self.shapes = []   # free-form shape string per record (tools/gpu_diag.py calls)
        self.layers = []   # model-YAML layer index per record (set by the graph walk: `layer`; -1 = outside a layer, e.g. NMS)
        self.layer = -1
        self.io = {}       # layer index -> (input elements, output elements) of the layer as the model graph defines it

    def start(self):
        self.on, self.records, self.shapes, self.layers, self.layer, self.io = True, [], [], [], -1, {}

    def note(self, rec, shape=""):
        """Append a finished record (callers that build their own event pairs)."""
        self.records.append(rec)
        self.shapes.append(shape)
        self.layers.append(self.layer)

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
        self.note((family, e0, e1, int(nbytes), int(flops)), shape)


TIMER = KernelTimer()
CONV_FAMILY = {0: "conv_igemm", 1: "conv1x1_ws", 2: "conv3x3_tile"}  # ymk_conv2d_last_variant()


def _tile(cout: int) -> str:
    """cout x pixel tile and wave grid of conv_igemm_kernel (csrc/conv.hip launch_conv / launch_conv_dual)."""
    return "128, 128, 2, 2" if cout > 64 else "64, 256, 1, 4" if cout > 32 else "32, 256, 1, 4" if cout > 16 else "16, 256, 1, 4"


def conv_kernel_name(variant: int, dtype, cin: int, cout: int, k: int, kpad: int, residual: bool, dual: bool = False) -> str:
    """Demangled name of the kernel instantiation a ymk_conv2d / ymk_conv1x1_cat2 call ran (as rocprofv3 prints it,
    with bf16_t = unsigned short): the timer tags conv calls with it so that bench.py's roofline object and the
    committed rocprof / PMC summaries refer to the same kernel."""
    t = "unsigned short" if dtype in H16 else "float"
    es = 2 if dtype in H16 else 4
    if dual and variant & 0xff != 3:
        return f"conv_igemm_kernel<{t}, {_tile(cout)}, 1, true>"
    if variant & 0xff == 3:   # LDS-DMA tiled core (default for bf16 3x3 with Cin >= 64; YMK_ENABLE bit 0: every shape); stages in bits 8+
        bn = ((variant >> 26) & 7) * 64 or (128 if cout % 128 == 0 else 64)   # cout-tile width / 64 in bits 26+, pixel-tile height in bits 16-25
        return f"conv_glds_kernel<{bn}, {(variant >> 8) & 0xff}, {((variant >> 16) & 0x3ff) or 256}>"
    if variant == 1:
        return f"conv1x1_ws_kernel<{t}, {kpad * es // 128}, {'true' if cout % 64 == 0 else 'false'}>"   # whole 64-cout groups: permuted rows
    if variant == 2:
        prefetch = residual and OPTIONS.res_prefetch
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


HAS_F16 = _lib.LIB_F16_PATH.exists()   # libymk_f16.so is built next to libymk.so by yolo_master_amd/build.py


def device_ok(t: torch.Tensor) -> bool:
    """True when libymk can take this tensor (the drop-in hooks fall through to the reference otherwise)."""
    return bool(t.is_cuda)


def _need_gpu(t: torch.Tensor) -> None:
    require_gpu(t)


def _nhwc(t: torch.Tensor):
    """Validate an NHWC view; return (B, H, W, C, ld)."""
    _need_gpu(t)
    if t.dtype in H16:
        _TLS.fmt = t.dtype
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
def conv2d(x, w_packed, bias, k: int, stride: int, act, out=None, residual=None, out_dtype=None, pool=False):
    """ymk_conv2d (include/ymk.h): y = act(conv(x) + bias) (+ residual).  act: False / True (SiLU), or "gelu" / "sigmoid" (no residual).
    pool: the caller's output feeds an ES-MoE router — where the pooled streaming 1x1 takes the shape (ymk_conv1x1_pool_chunks) the per-tile
    channel sums of the output are left in `out.gap_part` (fp32 [B, chunks, Cout]) and the router pools those instead of re-reading the map."""
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
    # act: False / True (SiLU) or "gelu" / "sigmoid" (ymk.h YMK_ACT_*: fused in the LDS-DMA core's epilogue, one in-place pass after the others)
    d = ConvDesc(DT[x.dtype], DT[out.dtype], B, H, W, Cin, Cout, k, stride, ldx, ldy, ldr, Kp,
                 _ACT[act] if isinstance(act, str) else (_lib.ACT_SILU if act else _lib.ACT_NONE))
    # pooled sums for an ES-MoE router: whenever the map is a whole number of 128-pixel tiles — from the convolution's own epilogue where the
    # streaming kernel takes the shape, else by ymk_pool_tiles128 in the SAME summation order (a routing decision must not depend on the batch size)
    want_pool = pool and OPTIONS.pooled_producers and not isinstance(act, str) and k == 1 and stride == 1 and (Ho * Wo) % 128 == 0 \
        and x.dtype in H16 and out.dtype == x.dtype
    chunks = int(lib.ymk_conv1x1_pool_chunks(C.byref(d))) if want_pool else 0
    e0 = TIMER.begin()
    if chunks > 0:
        part = torch.empty((B, chunks, Cout), dtype=torch.float32, device=x.device)
        check(lib.ymk_conv1x1_pooled(C.byref(d), _p(x), _p(w_packed), _p(bias), _p(residual), _p(out), _p(part), _stream()), "conv1x1_pooled")
        out.gap_part = part
    else:
        check(lib.ymk_conv2d(C.byref(d), _p(x), _p(w_packed), _p(bias), _p(residual), _p(out), _stream()), "conv2d")
    if e0 is not None:
        es = x.element_size()
        nbytes = (B * H * W * Cin + Cout * k * k * Cin + (B * Ho * Wo * Cout if residual is not None else 0)) * es \
            + B * Ho * Wo * Cout * out.element_size()
        name = conv_kernel_name(lib.ymk_conv2d_last_variant(), x.dtype, Cin, Cout, k, Kp, residual is not None)
        TIMER.end(e0, name, nbytes, 2 * B * Ho * Wo * Cout * k * k * Cin,
                  f"{Cin}->{Cout} k{k} s{stride} @{Ho}x{Wo}{' +res' if residual is not None else ''}")
    if want_pool and chunks == 0:
        part = torch.empty((B, Ho * Wo // 128, Cout), dtype=torch.float32, device=x.device)
        e0 = TIMER.begin()
        check(lib.ymk_pool_tiles128(DT[out.dtype], _p(out), ldy, B, Ho * Wo, Cout, _p(part), _stream()), "pool_tiles128")
        TIMER.end(e0, "moe_route", B * Ho * Wo * Cout * out.element_size(), B * Ho * Wo * Cout, f"C{Cout} @{Ho}x{Wo} tile sums")
        out.gap_part = part
    elif not want_pool and getattr(out, "gap_part", None) is not None:
        out.gap_part = None   # a caller-owned output tensor written before by a pooling producer: its sums describe the OLD contents
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
    TIMER.end(e0, conv_kernel_name(lib.ymk_conv2d_last_variant(), x2.dtype, C1 + C2, Cout, 1, Kp, False, dual=True),
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


def stem_pair_supported(dtype, cin: int, c0: int, c1: int, k0: int, s0: int, k1: int, s1: int) -> bool:
    """YMK_DISABLE bit 2048 switches the fused stem + row-1 kernel off (-> conv2d_stem + conv2d) for A/B runs."""
    return dtype in DT and bool(lib.ymk_stem_pair_supported(DT[dtype], cin, c0, c1, k0, s0, k1, s1)) and \
        OPTIONS.fused_stem_pair


def stem_pair(x_nchw, wt0, b0, w1, b1, out=None):
    """Layer 0 (stem 3x3/s2 + SiLU) and layer 1 (3x3/s2 + SiLU) as one kernel (include/ymk.h ymk_stem_pair): x fp32 NCHW, wt0 the
    stem's transposed fp32 weights [27][C0], w1 the next convolution's packed bf16 weights; returns layer 1's NHWC bf16 output."""
    _need_gpu(x_nchw)
    x_nchw = x_nchw.contiguous().float()
    B, _, H, W = x_nchw.shape
    C0, C1 = wt0.shape[1], w1.shape[0]
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    if out is None:
        out = new_act(B, H2, W2, C1, w1.dtype, x_nchw.device)
    use_format(w1.dtype)
    ldy = _nhwc(out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_stem_pair(_p(x_nchw), B, H, W, _p(wt0), _p(b0), C0, _p(w1), w1.shape[1], _p(b1), C1, _p(out), ldy, _stream()), "stem_pair")
    TIMER.end(e0, "stem_pair", B * 3 * H * W * 4 + B * H2 * W2 * C1 * 2, 2 * B * (H1 * W1 * C0 * 27 + H2 * W2 * C1 * 9 * C0), f"3->{C0}->{C1} @{H}x{W}")
    return out


def c3k2_fused_supported(dtype, c1: int, c2: int, c: int, n: int, c3k: bool, shortcut: bool) -> bool:
    """YMK_DISABLE bit 8192 switches the fused C3k2 block off (-> its four convolutions) for A/B runs."""
    return dtype in DT and bool(lib.ymk_c3k2_fused_supported(DT[dtype], c1, c2, c, n, int(bool(c3k)), int(bool(shortcut)))) and \
        OPTIONS.fused_c3k2


def c3k2_fused(x, p1, pa, pb, p2, out=None, pool=True):
    """C3k2 (c3k = False, n = 1, c = 32) as one kernel (include/ymk.h ymk_c3k2_fused_pooled).  p1 / pa / pb / p2 = (packed bf16 weights,
    fp32 bias) of cv1, m[0].cv1, m[0].cv2, cv2; x / out NHWC bf16 views.  pool: the kernel also leaves the per-tile channel sums of its
    output in `out.gap_part` (fp32 [B, chunks, 128]) — an ES-MoE router consuming `out` pools those instead of re-reading the map."""
    B, H, W, Cin, ldx = _nhwc(x)
    Cout = p2[0].shape[0]
    if out is None:
        out = new_act(B, H, W, Cout, x.dtype, x.device)
    ldy = _nhwc(out)[4]
    part = torch.empty((B, lib.ymk_c3k2_fused_pool_chunks(H, W), Cout), dtype=torch.float32, device=x.device) if pool else None
    e0 = TIMER.begin()
    check(lib.ymk_c3k2_fused_pooled(_p(x), ldx, B, H, W, _p(p1[0]), p1[0].shape[1], _p(p1[1]), _p(pa[0]), pa[0].shape[1], _p(pa[1]), _p(pb[0]),
                                    pb[0].shape[1], _p(pb[1]), _p(p2[0]), p2[0].shape[1], _p(p2[1]), _p(out), ldy, _p(part), None, _stream()),
          "c3k2_fused")
    TIMER.end(e0, "c3k2_fused", B * H * W * (Cin + Cout) * 2, 2 * B * H * W * (64 * 64 + 288 * 16 + 144 * 32 + 96 * 128), f"{Cin}->{Cout} @{H}x{W}")
    if pool:
        out.gap_part = part
    elif getattr(out, "gap_part", None) is not None:
        out.gap_part = None   # (see conv2d)
    return out


def detect_cls_fused_supported(dtype, cin: int, c3: int, nc: int) -> bool:
    """YMK_DISABLE bit 16384 switches the fused Detect class branch off (-> its five convolutions) for A/B runs."""
    return dtype in DT and bool(lib.ymk_detect_cls_fused_supported(DT[dtype], cin, c3, nc)) and \
        OPTIONS.fused_detect_cls


def detect_cls_fused(x, d1, p1, d2, p2, w3, out=None, y=None, nc=0, a_off=0, raw=True):
    """One pyramid level's Detect class branch as one kernel (include/ymk.h ymk_detect_cls_fused).  d1 / d2 = (packed depthwise weights
    [9][C], fp32 bias), p1 / p2 = (packed 1x1 weights, fp32 bias), w3 = (packed [ncpad][Kpad], fp32 bias); returns fp32 [B, H, W, ncpad].
    y: the Detect output fp32 [B, 4+nc, A] (with `y.best`, ops.detect_decode) -> the kernel also writes sigmoid(logits) into its class
    rows at anchors a_off.. and the per-anchor best class: the class half of detect_decode.  raw=False then skips the logits (returns None)."""
    B, H, W, Cin, ldx = _nhwc(x)
    ncpad = w3[0].shape[0]
    if y is None:
        raw = True
    if raw and out is None:
        out = torch.empty((B, H, W, ncpad), dtype=torch.float32, device=x.device)
    ldy = _nhwc(out)[4] if raw else 0
    bc = bi = None
    A = 0
    if y is not None:
        A = y.shape[2]
        assert y.dtype == torch.float32 and y.is_contiguous() and y.shape[0] == B and y.shape[1] == 4 + nc and a_off + H * W <= A
        best = getattr(y, "best", None)
        if best is not None:
            bc, bi = best[0], best[1]
            assert bc.dtype == torch.float32 and bi.dtype == torch.int32 and bc.is_contiguous() and bi.is_contiguous() and \
                tuple(bc.shape) == tuple(bi.shape) == (B, A)
    e0 = TIMER.begin()
    check(lib.ymk_detect_cls_fused(_p(x), ldx, B, H, W, Cin, _p(d1[0]), _p(d1[1]), _p(p1[0]), p1[0].shape[1], _p(p1[1]), _p(d2[0]), _p(d2[1]),
                                   _p(p2[0]), p2[0].shape[1], _p(p2[1]), _p(w3[0]), w3[0].shape[1], _p(w3[1]), ncpad, _p(out) if raw else None, ldy,
                                   _p(y), int(nc), int(a_off), A, _p(bc), _p(bi), _stream()),
          "detect_cls_fused")
    TIMER.end(e0, "detect_cls_fused", B * H * W * (Cin * 2 + (ncpad * 4 if raw else 0) + ((nc * 4 + 8) if y is not None else 0)),
              2 * B * H * W * (9 * Cin + Cin * 128 + 9 * 128 + 128 * 128 + 128 * ncpad), f"{Cin}->128->{ncpad} @{H}x{W}" + (" +decode" if y is not None else ""))
    return out if raw else None


def detect_box_tail_supported(dtype, cin: int, reg_max: int, nc: int) -> bool:
    """YMK_DISABLE bit 4194304 switches the fused decode of the Detect head off (-> 1x1 convolution, fp32 logits, detect_decode)."""
    return dtype in DT and bool(lib.ymk_detect_box_tail_supported(DT[dtype], cin, reg_max, nc)) and \
        OPTIONS.fused_decode


def detect_box_tail(x, w_packed, bias, y, stride: float, a_off: int, reg_max: int, raw: bool = False):
    """The last 1x1 of a level's Detect box branch + DFL + dist2bbox -> rows 0..3 of y (include/ymk.h ymk_detect_box_tail).
    x: [B, H, W, 64] 16-bit; y fp32 [B, 4+nc, A].  raw=True also returns the fp32 box logits [B, H, W, 64] (else None)."""
    B, H, W, Cin, ldx = _nhwc(x)
    assert y.dtype == torch.float32 and y.is_contiguous() and y.shape[0] == B
    out = torch.empty((B, H, W, 4 * reg_max), dtype=torch.float32, device=x.device) if raw else None
    e0 = TIMER.begin()
    check(lib.ymk_detect_box_tail(DT[x.dtype], _p(x), ldx, B, H, W, _p(w_packed), w_packed.shape[1], _p(bias), reg_max, y.shape[1] - 4,
                                  float(stride), a_off, y.shape[2], _p(y), _p(out), _stream()), "detect_box_tail")
    TIMER.end(e0, "detect_box_tail", B * H * W * (Cin * 2 + 16 + (256 if raw else 0)), 2 * B * H * W * Cin * 64, f"{Cin}->64 +dfl @{H}x{W}")
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
    """Router + dispatch decision + CSR + eval-time state (include/ymk.h ymk_esmoe_route).  thr < 0 = dense forward.
    Returns (route_w [B,E], gate_w [B,E], sel [B,top_k], csr_off [E+1], csr_pair [B*top_k], state [E+1])."""
    B, H, W, Cc, ldx = _nhwc(x)
    hidden, E = w1.shape[0], w2.shape[0]
    dev = x.device
    route_w = torch.empty((B, E), dtype=torch.float32, device=dev)
    gate_w = torch.empty((B, E), dtype=torch.float32, device=dev)
    sel = torch.empty((B, top_k), dtype=torch.int32, device=dev)
    csr_off = torch.empty((E + 1,), dtype=torch.int32, device=dev)
    csr_pair = torch.empty((B * top_k,), dtype=torch.int32, device=dev)
    state = torch.empty((E + 1,), dtype=torch.float32, device=dev)
    part = getattr(x, "gap_part", None)   # per-chunk channel sums left by the kernel that produced x (c3k2_fused)
    e0 = TIMER.begin()
    if part is not None and part.dtype == torch.float32 and part.dim() == 3 and part.shape[0] == B and part.shape[2] == Cc and part.is_contiguous():
        check(lib.ymk_esmoe_route_pooled(_p(part), part.shape[1], B, H, W, Cc, _p(w1), _p(b1), _p(w2), _p(b2), hidden, E, top_k, float(thr),
                                         _p(route_w), _p(gate_w), _p(sel), _p(csr_off), _p(csr_pair), _p(state), _p(flags), _stream()),
              "esmoe_route_pooled")
        TIMER.end(e0, "moe_route", part.numel() * 4, B * H * W * Cc, f"C{Cc} @{H}x{W} pooled")
    else:
        nbytes = lib.ymk_esmoe_route_workspace_bytes(B, Cc, H, W)
        ws = torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=dev)
        check(lib.ymk_esmoe_route(DT[x.dtype], _p(x), B, H, W, Cc, ldx, _p(w1), _p(b1), _p(w2), _p(b2), hidden, E, top_k,
                                  float(thr), _p(route_w), _p(gate_w), _p(sel), _p(csr_off), _p(csr_pair), _p(state),
                                  _p(flags), _p(ws), nbytes, _stream()), "esmoe_route")
        TIMER.end(e0, "moe_route", B * H * W * Cc * x.element_size(), B * H * W * Cc, f"C{Cc} @{H}x{W}")
    return route_w, gate_w, sel, csr_off, csr_pair, state


def esmoe_dw(x, dw_w, dw_off, ksizes, kmax: int, top_k: int, sel, csr_off, csr_pair):
    """Depthwise stage of the retained (image, expert) pairs: the LDS-tiled VALU stencil per pair (csrc/dwconv.hip)."""
    B, H, W, Cc, ldx = _nhwc(x)
    E = ksizes.numel()
    out = torch.empty((B * top_k, H, W, Cc), dtype=x.dtype, device=x.device)
    e0 = TIMER.begin()
    check(lib.ymk_esmoe_dw(DT[x.dtype], _p(x), B, H, W, Cc, ldx, _p(dw_w), _p(dw_off), _p(ksizes), E, top_k, kmax, _p(sel),
                           _p(csr_off), _p(csr_pair), _p(out), _stream()), "esmoe_dw")
    if e0 is not None:  # algorithmic traffic: every image read once, one plane written per retained (image, expert) pair
        e1 = TIMER.begin()
        live = (sel >= 0).reshape(-1)      # (from the selection of THIS call's images: the layer may be walked in image chunks)
        npairs = int(live.sum())
        k2 = int((ksizes.long()[sel.reshape(-1).clamp_min(0).long()] ** 2 * live).sum())
        TIMER.note(("moe_dw", e0, e1, (B + npairs) * H * W * Cc * x.element_size(), 2 * k2 * H * W * Cc), f"C{Cc} @{H}x{W} pairs {npairs}")
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
        TIMER.note(("moe_pw", e0, e1, (npairs * H * W * Cc + E * Cout * Cc + B * H * W * Cout) * es, 2 * npairs * H * W * Cc * Cout),
                   f"{Cc}->{Cout} @{H}x{W} pairs {npairs}")
    return out


def mlp_fused_supported(dtype, C: int, hidden: int) -> bool:
    """YMK_DISABLE bit 1024 switches the fused ABlock MLP off (-> two 1x1 convolutions) for A/B runs."""
    return dtype in DT and bool(lib.ymk_mlp_fused_supported(DT[dtype], C, hidden)) and OPTIONS.fused_mlp


def mlp_fused(x, w1, b1, w2, b2, out=None):
    """y = x + W2 * SiLU(W1 * x + b1) + b2 per token (ABlock's x + mlp(x)), one kernel: x / y NHWC bf16 views, w1 [hidden][Kpad],
    w2 [C][Kpad] packed bf16 (BN folded), fp32 biases."""
    B, H, W, Cc, ldx = _nhwc(x)
    hidden = w1.shape[0]
    if out is None:
        out = new_act(B, H, W, Cc, x.dtype, x.device)
    ldy = _nhwc(out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_mlp_fused(_p(x), ldx, _p(w1), w1.shape[1], _p(b1), _p(w2), w2.shape[1], _p(b2), _p(out), ldy, B * H * W, Cc, hidden,
                            _stream()), "mlp_fused")
    TIMER.end(e0, "mlp_fused", (2 * B * H * W * Cc + 2 * Cc * hidden) * x.element_size(), 4 * B * H * W * Cc * hidden, f"{Cc}->{hidden}->{Cc} @{H}x{W}")
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


def area_attn_qkv_supported(dtype, C: int, heads: int, N: int, area: int) -> bool:
    """The qkv projection inside the attention kernel (csrc/attn.hip area_attn_qkv_kernel): 16-bit, C = heads * 32 in {64, 128}, at most
    512 tokens per area.  OPTIONS.fused_qkv_attn / YMK_DISABLE bit 2097152 switch it off (-> 1x1 convolution + area_attn)."""
    return dtype in DT and OPTIONS.fused_qkv_attn and bool(lib.ymk_area_attn_qkv_supported(DT[dtype], C, heads, N, area))


def area_attn_qkv(x, w, b, heads: int, area: int, out=None, v_out=None):
    """attention(q, k, v) with qkv = w x + b computed inside the kernel (include/ymk.h ymk_area_attn_qkv): x NHWC [B, H, W, C], w the packed
    [3C][Kpad] weights with rows [Q | K | V]; returns (attention output, v), both [B, H, W, C]."""
    B, H, W, Cc, ldx = _nhwc(x)
    if out is None:
        out = new_act(B, H, W, Cc, x.dtype, x.device)
    if v_out is None:
        v_out = new_act(B, H, W, Cc, x.dtype, x.device)
    ldo, ldv = _nhwc(out)[4], _nhwc(v_out)[4]
    e0 = TIMER.begin()
    check(lib.ymk_area_attn_qkv(DT[x.dtype], _p(x), ldx, _p(w), w.shape[1], _p(b), _p(out), ldo, _p(v_out), ldv, B, H * W, Cc, heads, area,
                                _stream()), "area_attn_qkv")
    es = x.element_size()
    TIMER.end(e0, "area_attn_qkv", 3 * B * H * W * Cc * es + w.numel() * es, 2 * B * H * W * Cc * 3 * Cc + 4 * B * H * W * (H * W // area) * Cc,
              f"C{Cc} heads {heads} area {area} @{H}x{W}")
    return out, v_out


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
def detect_decode(box_l, cls_l, y, stride: float, a_off: int, reg_max: int, best=None):
    """cls_l: fp32 [B, Hl, Wl, nc], dense or a view of rows padded to 4*ceil(nc/4) channels (tail conv with nc % 4 != 0).
    best: optional (conf fp32 [B, A], cls int32 [B, A]) filled with every anchor's largest class score and its class — nms_batched
    takes them (`y.best`) instead of reading the class rows of y again."""
    B, Hl, Wl, _, _ = _nhwc(box_l)
    nc = cls_l.shape[-1]
    ldc = _nhwc(cls_l)[4]
    assert box_l.is_contiguous() and box_l.dtype == torch.float32 and cls_l.dtype == torch.float32 and y.shape[1] == 4 + nc
    e0 = TIMER.begin()
    bc, bi = (best[0], best[1]) if best is not None else (None, None)
    if best is not None:
        assert bc.dtype == torch.float32 and bi.dtype == torch.int32 and bc.is_contiguous() and bi.is_contiguous() and \
            tuple(bc.shape) == tuple(bi.shape) == (B, y.shape[2])
    check(lib.ymk_detect_decode(_p(box_l), _p(cls_l), _p(y), B, Hl, Wl, reg_max, nc, ldc, float(stride), a_off, y.shape[2],
                                _p(bc), _p(bi), _stream()), "detect_decode")
    TIMER.end(e0, "detect_decode", B * Hl * Wl * (4 * reg_max + nc + 4 + nc) * 4, B * Hl * Wl * (4 * reg_max * 4 + nc * 4))
    return y


_STATUS = {}


def _zero_status(dev):
    """The NMS status word: reserved since any candidate count is selected on the device (include/ymk.h YMK_FLAG_NMS_OVERFLOW); one
    zero per device, never written."""
    t = _STATUS.get(dev)
    if t is None:
        t = _STATUS[dev] = torch.zeros((1,), dtype=torch.int32, device=dev)
    return t


def nms_pack_numel(B: int, max_det: int) -> int:
    """32-bit words of one packed NMS result: dets [B, max_det, 6] f32 | idx [B, max_det] i32 | counts [B] i32 (one allocation, so that
    a multi-GPU step gathers its results with ONE collective, yolo_master_amd/dist.py gather_packed)."""
    return B * max_det * 7 + B


def nms_pack_views(pack: torch.Tensor, B: int, max_det: int):
    """(dets, counts, idx) views of a packed result buffer (float32 [..., nms_pack_numel]); leading dims are kept."""
    lead = pack.shape[:-1]
    n6, n1 = B * max_det * 6, B * max_det
    dets = pack[..., :n6].reshape(*lead, B, max_det, 6)
    idx = pack[..., n6:n6 + n1].view(torch.int32).reshape(*lead, B, max_det)
    counts = pack[..., n6 + n1:n6 + n1 + B].view(torch.int32).reshape(*lead, B)
    return dets, counts, idx


def nms_batched(y, conf: float, iou: float, multi_label: bool, agnostic: bool, max_det: int, max_nms: int,
                max_wh: float, cw_sigma: float | None = None, cw_pool: int = 3000, class_keep: torch.Tensor | None = None,
                pack: torch.Tensor | None = None, nc: int = 0, use_best: bool = True):
    """Returns (dets [B,max_det,6], counts [B] int32, idx [B,max_det] int32, status [1] int32).
    class_keep: uint8 [nc] on the GPU (the `classes=` filter, utils/nms.py:63,132) or None.
    pack: optional contiguous float32 [nms_pack_numel(B, max_det)] buffer the three outputs are carved from.
    nc: number of class rows when y carries extra rows behind them (utils/nms.py:76-81: a Segment head's mask coefficients);
    0 = every row behind the box is a class.  The carried rows of the kept detections: nms_gather_rows.
    use_best: take the producer's per-anchor best class (`y.best`, attached by Detect) instead of a pass over the class rows when it is
    valid for y.  CONTRACT ("y.best"): valid means attached to this very tensor and y's autograd version counter unchanged
    since Detect stamped it — writes that bypass the counter (`y.data`, raw-pointer kernels, graph replays into the same buffer WITHOUT the
    producer) are invisible to that check; a caller that edits y that way passes use_best=False (or deletes `y.best`)."""
    _need_gpu(y)
    assert y.dtype == torch.float32 and y.is_contiguous()
    B, ch, A = y.shape
    nc = int(nc) or ch - 4
    extra = ch - 4 - nc
    assert extra >= 0, f"nc = {nc} but the prediction has {ch - 4} rows behind the box"
    dev = y.device
    nbytes = lib.ymk_nms_workspace_bytes(B, nc, A, int(multi_label), max_nms)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    # outputs are written in full by the kernels (rows past the count zeroed by nms_greedy_kernel): no fill kernels in the step
    if pack is not None:
        assert pack.dtype == torch.float32 and pack.is_contiguous() and pack.numel() == nms_pack_numel(B, max_det) and pack.device == dev
        dets, counts, idx = nms_pack_views(pack, B, max_det)
    else:
        dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        idx = torch.empty((B, max_det), dtype=torch.int32, device=dev)
    status = _zero_status(dev)
    e0 = TIMER.begin()
    if class_keep is not None:
        assert class_keep.dtype == torch.uint8 and class_keep.numel() == nc and class_keep.device == y.device and class_keep.is_contiguous()
    # the producer's per-anchor best class (Detect attaches it to the y it returns: `y.best`), valid for exactly this tensor
    # ... and only while y is untouched since (Detect.finish stamps y's version counter: an in-place edit by the caller drops it)
    best = getattr(y, "best", None)
    bc, bi = (None, None)
    if use_best and best is not None and not multi_label and extra == 0 and tuple(best[0].shape) == (B, A) and best[0].device == dev and \
            len(best) == 3 and not y.is_inference() and best[2] == y._version:
        bc, bi = best[0], best[1]
    check(lib.ymk_nms_batched(_p(y), B, nc, extra, A, float(conf), float(iou), int(multi_label), int(agnostic), max_det, max_nms,
                              float(max_wh), _p(class_keep), _p(bc), _p(bi), _p(dets), _p(counts), _p(idx), _p(status), _p(ws), nbytes,
                              _stream()), "nms_batched")
    if cw_sigma is not None:
        check(lib.ymk_cw_refine(B, nc, A, int(multi_label), int(agnostic), max_nms, max_det, float(iou), float(cw_sigma), cw_pool,
                                _p(dets), _p(counts), _p(ws), nbytes, _stream()), "cw_refine")
    TIMER.end(e0, "nms", B * ((4 + nc) if bc is None else 6) * A * 4 + B * max_det * 28, B * (4 + nc) * A)
    return dets, counts, idx, status


def nms_gather_rows(y, nc: int, idx, counts, out=None):
    """The rows of y behind the class rows, for the detections NMS kept (the `mask` columns of the reference's output rows,
    utils/nms.py:117,122,127): y fp32 [B, 4+nc+extra, A], idx int32 [B, max_det], counts int32 [B] (nms_batched's outputs)
    -> fp32 [B, max_det, extra], zeros behind each image's count."""
    _need_gpu(y)
    B, ch, A = y.shape
    extra, max_det = ch - 4 - nc, idx.shape[1]
    assert extra > 0 and y.dtype == torch.float32 and y.is_contiguous() and idx.dtype == torch.int32 and idx.is_contiguous()
    if out is None:
        out = torch.empty((B, max_det, extra), dtype=torch.float32, device=y.device)
    check(lib.ymk_nms_gather_rows(_p(y), B, ch, A, 4 + nc, extra, _p(idx), _p(counts), max_det, _p(out), _stream()), "nms_gather_rows")
    return out


# ============================================================================= config-5 rows (MoA / MoT / gated MoE)
# Entry points of include/ymk_mixture.h.  Their CONTRACT is fixed here — it is what nn/mixture.py is written against and
# what tests/emu_ops.py restates for the CPU-side host tests.  The HIP kernels (csrc/mixture.hip, csrc/mixattn.hip) are
# validated on MI355X by tests/test_gpu_mixture.py (first hardware run: round 2, profiles/r02_first_hw_run.log).  There is
# no CPU / PyTorch fallback: every wrapper goes through `_nhwc` -> `require_gpu`.  Conventions as above: NHWC views
# [B, H, W, C] with a pixel stride ld >= C, activations in the compute dtype, statistics / gates / router math fp32.
def _tensor_bytes(obj) -> int:
    if torch.is_tensor(obj):
        return obj.numel() * obj.element_size()
    if isinstance(obj, (list, tuple)):
        return sum(_tensor_bytes(o) for o in obj)
    return 0


def _timed(family: str, flops=None):
    """Per-call HIP-event timing of a config-5 op under `family` (bench.py's roofline leg; inactive unless TIMER.start() was called).
    Algorithmic bytes = every tensor argument once + the result once; flops from the op's shapes where it has a contraction."""
    import functools

    def wrap(fn):
        @functools.wraps(fn)
        def run(*a, **kw):
            e0 = TIMER.begin()
            r = fn(*a, **kw)
            if e0 is not None:
                outs = kw.get("out")
                nb = _tensor_bytes(a) + _tensor_bytes([v for k, v in kw.items() if k != "out"]) + (_tensor_bytes(r) if outs is None else _tensor_bytes(outs))
                TIMER.end(e0, family, nb, int(flops(*a, **kw)) if flops else 0)
            return r
        return run
    return wrap


def _attn_flops(q, k, v, heads, hd, *a, **kw):
    nq, nk = q.shape[0] * q.shape[1] * q.shape[2], k.shape[1] * k.shape[2]
    return 4 * nq * nk * heads * hd


def _win_flops(q, k, v, heads, hd, scale, win, *a, **kw):
    return 4 * q.shape[0] * q.shape[1] * q.shape[2] * win * win * heads * hd


def _lin_flops(q, k, v, rf, heads, hd, *a, **kw):
    return 8 * q.shape[0] * q.shape[1] * q.shape[2] * rf.shape[0] * heads * hd   # phi(k), phi(k)^T v, phi(q), phi(q) kv


ACT_CODES = (False, True, "silu", "sigmoid", "gelu")   # conv / norm epilogues of the config-5 modules
_ACT = {False: _lib.ACT_NONE, None: _lib.ACT_NONE, True: _lib.ACT_SILU, "silu": _lib.ACT_SILU, "sigmoid": _lib.ACT_SIGMOID,
        "gelu": _lib.ACT_GELU}


def _out_like(x, out, dtype=None, shape=None):
    if out is None:
        out = torch.empty(tuple(shape or x.shape), dtype=dtype or x.dtype, device=x.device)
    return out, _nhwc(out)[4]


def conv2d_act(x, w_packed, bias, k: int, stride: int, act, out=None, residual=None, out_dtype=None):
    """ymk_conv2d with the extended epilogue set: act in ACT_CODES ("gelu" = exact erf GELU, nn.GELU default); sigmoid and GELU are
    applied in the LDS-DMA convolution core's epilogue, or by one in-place pass inside ymk_conv2d where another core takes the shape."""
    if act in (False, True, "silu"):
        return conv2d(x, w_packed, bias, k, stride, bool(act), out=out, residual=residual, out_dtype=out_dtype)
    if act not in ACT_CODES:
        raise ValueError(f"unknown activation {act!r}")
    if residual is not None:
        raise NotImplementedError("conv2d_act: sigmoid / gelu epilogues take no residual")
    return conv2d(x, w_packed, bias, k, stride, act, out=out, out_dtype=out_dtype)


@_timed("group_norm")
def group_norm(x, groups: int, weight, bias, eps: float, act=False, out=None, out_dtype=None, affine_rows=None, residual=None):
    """GroupNorm over (H, W, C/groups) per image and group, biased variance, fp32 statistics (torch.nn.GroupNorm;
    safe group counts nn/modules/utils.py:108-115).  weight/bias fp32 [C], or None (no affine), or [R][C] with
    affine_rows int32 [B] choosing the row per image (FusedExpertGroup's per-expert affine, moe/gated.py:1058-1090).
    act in (False, "silu"); residual (same shape) is added after the activation (MoTBlock's out_norm(.) + x,
    mot/block.py:413-417).  x may be a channel slice (C channels of a wider buffer); any C >= 1; out may alias x."""
    B, H, W, Cc, ldx = _nhwc(x)
    out, ldy = _out_like(x, out, out_dtype)
    ldr = _nhwc(residual)[4] if residual is not None else 0
    if residual is not None and residual.dtype != out.dtype:
        raise ValueError("group_norm: the residual has the output's dtype")
    ws = torch.empty((B * groups * (2 + 3 * 256),), dtype=torch.float32, device=x.device)   # (mean, rstd) + 256 chunk partials (mean, M2, n) per slab
    check(lib.ymk_group_norm(DT[x.dtype], _p(x), ldx, _p(out), DT[out.dtype], ldy, _p(residual), ldr, B, H * W, Cc, groups,
                             _p(weight), _p(bias), _p(affine_rows), float(eps), _ACT[act], _p(ws), _stream()), "group_norm")
    return out


@_timed("layer_norm")
def layer_norm(x, weight, bias, eps: float, out=None):
    """LayerNorm over the channel vector of every token (torch.nn.LayerNorm(C)), fp32 statistics."""
    B, H, W, Cc, ldx = _nhwc(x)
    out, ldy = _out_like(x, out)
    check(lib.ymk_layer_norm(DT[x.dtype], _p(x), ldx, _p(out), ldy, B * H * W, Cc, _p(weight), _p(bias), float(eps), _stream()),
          "layer_norm")
    return out


@_timed("eltwise")
def _eltwise(op, a, b, alpha, out, what):
    B, H, W, Cc, lda = _nhwc(a)
    ldb = _nhwc(b)[4]
    if a.dtype != b.dtype or a.shape != b.shape:
        raise ValueError(f"{what}: operands must agree in shape and dtype")
    out, ldy = _out_like(a, out)
    check(lib.ymk_eltwise(op, DT[a.dtype], _p(a), lda, _p(b), ldb, _p(out), ldy, B * H * W, Cc, float(alpha), _stream()), what)
    return out


def eltwise_mul(a, b, out=None, act_a=None):
    """out = act_a(a) * b, same shapes; act_a in (None, "sigmoid") (GLU of the MoT local expert: sigmoid(gate) * value,
    mot/experts.py:160-166)."""
    if act_a not in (None, "sigmoid"):
        raise ValueError(act_a)
    return _eltwise(_lib.ELT_SIGMOID_MUL if act_a else _lib.ELT_MUL, a, b, 0.0, out, "eltwise_mul")


def clamp_add(a, b, limit: float, out=None):
    """out = clamp(a, -limit, limit) + b (UltraOptimizedMoE: `shared + expert_output.clamp_(-1e4, 1e4)`, moe/utils.py:203, modules.py:224)."""
    return _eltwise(_lib.ELT_CLAMP_ADD, a, b, limit, out, "clamp_add")


def lerp(a, b, alpha: float, out=None):
    """out = (1 - alpha) * a + alpha * b (exact / linear attention blend, moa/heads.py:366-374)."""
    return _eltwise(_lib.ELT_LERP, a, b, alpha, out, "lerp")


@_timed("eltwise")
def fma_gate(x, a, b, scale, out=None):
    """out = x + scale * a * b with scale a host float; b is a map [B,H,W,C] or a per-image channel gate [B,1,1,C] fp32
    (detail gate x*(1+s*g), context mixer x+s*c*gate, refinement x+s*r*g: moe/gated.py:1171-1218, hooks.py:60-68)."""
    B, H, W, Cc, ldx = _nhwc(x)
    lda = _nhwc(a)[4]
    per_image = tuple(b.shape) == (B, 1, 1, Cc) and (H, W) != (1, 1)
    if per_image:
        if b.dtype != torch.float32 or not b.is_contiguous():
            raise ValueError("fma_gate: a per-image gate is a contiguous fp32 [B,1,1,C] tensor")
        ldb = Cc
    else:
        ldb = _nhwc(b)[4]
    out, ldy = _out_like(x, out)
    check(lib.ymk_fma_gate(DT[x.dtype], _p(x), ldx, _p(a), lda, _p(b), DT[b.dtype], ldb, int(per_image), float(scale), _p(out), ldy,
                           B, H * W, Cc, _stream()), "fma_gate")
    return out


@_timed("eltwise")
def channel_gate(x, gate, out=None):
    """out = x * gate with gate fp32 [B,1,1,C] (squeeze-excite gate of the gated MoE, moe/gated.py:333-341)."""
    B, H, W, Cc, ldx = _nhwc(x)
    if gate.dtype != torch.float32 or tuple(gate.shape) != (B, 1, 1, Cc) or not gate.is_contiguous():
        raise ValueError("channel_gate: gate is a contiguous fp32 [B,1,1,C] tensor")
    out, ldy = _out_like(x, out)
    check(lib.ymk_channel_gate(DT[x.dtype], _p(x), ldx, _p(gate), _p(out), ldy, B, H * W, Cc, _stream()), "channel_gate")
    return out


@_timed("weighted_sum")
def weighted_sum(weights, parts, out=None):
    """out = sum_e weights[..., e] * parts[e]; weights fp32 [B,H,W,>=E] per token or [B,1,1,>=E] per image
    (MoA head mix moa/block.py:230-262, MoT expert blend mot/block.py:360-417, gated expert mix)."""
    B, H, W, Cc, ldp = _nhwc(parts[0])
    E = len(parts)
    if not 1 <= E <= 4 or any(_nhwc(p_)[4] != ldp or p_.dtype != parts[0].dtype or p_.shape != parts[0].shape for p_ in parts):
        raise ValueError("weighted_sum: 1..4 parts of one shape, dtype and pixel stride")
    per_image = tuple(weights.shape[:3]) == (B, 1, 1) and (H, W) != (1, 1)
    if weights.dtype != torch.float32 or weights.stride(3) != 1:
        raise ValueError("weighted_sum: fp32 weights")
    ldw = weights.stride(0) if per_image else _nhwc(weights)[4]
    out, ldy = _out_like(parts[0], out)
    ptrs = [_p(p_) for p_ in parts] + [None] * (4 - E)
    check(lib.ymk_weighted_sum(DT[parts[0].dtype], _p(weights), ldw, int(per_image), E, *ptrs, ldp, _p(out), ldy, B, H * W, Cc,
                               _stream()), "weighted_sum")
    return out


@_timed("pool")
def mean_upsampled(parts, out=None):
    """out = mean_i nearest_resize(parts[i] -> size of parts[0]) (F.interpolate mode="nearest": src = floor(dst * h / H));
    PyramidContextMixer.forward moe/gated.py:1209-1216."""
    B, H, W, Cc, _ = _nhwc(parts[0])
    n = len(parts)
    if not 1 <= n <= 4:
        raise ValueError("mean_upsampled: 1..4 parts")
    geo = [_nhwc(p_) for p_ in parts]
    arr = lambda vals: (C.c_int32 * n)(*vals)   # noqa: E731
    out, ldy = _out_like(parts[0], out)
    ptrs = [_p(p_) for p_ in parts] + [None] * (4 - n)
    check(lib.ymk_mean_upsampled(DT[parts[0].dtype], n, *ptrs, arr([g[1] for g in geo]), arr([g[2] for g in geo]),
                                 arr([g[4] for g in geo]), _p(out), ldy, B, H, W, Cc, _stream()), "mean_upsampled")
    return out


@_timed("pool")
def adaptive_avg_pool(x, Ho: int, Wo: int, out=None, out_dtype=None):
    """F.adaptive_avg_pool2d bins: rows [floor(i*H/Ho), ceil((i+1)*H/Ho))."""
    B, H, W, Cc, ldx = _nhwc(x)
    out, ldy = _out_like(x, out, out_dtype, (B, Ho, Wo, Cc))
    check(lib.ymk_adaptive_avg_pool(DT[x.dtype], _p(x), ldx, _p(out), DT[out.dtype], ldy, B, H, W, Cc, Ho, Wo, _stream()),
          "adaptive_avg_pool")
    return out


@_timed("pool")
def avg_pool(x, k: int, out=None, out_dtype=None):
    """F.avg_pool2d(kernel_size=k, stride=k): floor(H/k) x floor(W/k) outputs, remainder rows/columns dropped."""
    B, H, W, Cc, ldx = _nhwc(x)
    out, ldy = _out_like(x, out, out_dtype, (B, H // k, W // k, Cc))
    check(lib.ymk_avg_pool(DT[x.dtype], _p(x), ldx, _p(out), DT[out.dtype], ldy, B, H, W, Cc, k, _stream()), "avg_pool")
    return out


@_timed("pool")
def channel_stats(x, want_std: bool = False):
    """Per image and channel mean (and biased std) over H*W in fp32: returns [B,1,1,C] (or [B,1,1,2C] = [mean | std],
    DualStreamGateRouter's global stream moe/gated.py:133-139)."""
    B, H, W, Cc, ldx = _nhwc(x)
    out = torch.empty((B, 1, 1, 2 * Cc if want_std else Cc), dtype=torch.float32, device=x.device)
    nchunk = min(64, (H * W + 1023) // 1024)
    ws = torch.empty((B * nchunk * 2 * Cc,), dtype=torch.float32, device=x.device) if nchunk > 1 else None
    check(lib.ymk_channel_stats(DT[x.dtype], _p(x), ldx, _p(out), B, H * W, Cc, int(want_std), _p(ws), _stream()), "channel_stats")
    return out


def _qkv_geometry(q, k, v, heads, hd, what):
    B, Hq, Wq, Cq, ldq = _nhwc(q)
    Bk, Hk, Wk, Ck, ldk = _nhwc(k)
    ldv = _nhwc(v)[4]
    if Cq != heads * hd or Ck != Cq or v.shape != k.shape or Bk != B or not (q.dtype == k.dtype == v.dtype):
        raise ValueError(f"{what}: q [B,Hq,Wq,heads*hd], k / v [B,Hk,Wk,heads*hd] of one dtype")
    return B, Hq, Wq, Hk, Wk, ldq, ldk, ldv


@_timed("mix_attention", _attn_flops)
def attention(q, k, v, heads: int, hd: int, scale: float, out=None):
    """softmax(q k^T * scale) v per (image, head).  q [B,Hq,Wq,heads*hd], k/v [B,Hk,Wk,heads*hd] channel-slice views
    (tokens row-major); any hd that is a multiple of 8.  MoA regional / global-exact heads (moa/heads.py:208-253,
    354-365), MoT local expert (mot/experts.py:150-156)."""
    B, Hq, Wq, Hk, Wk, ldq, ldk, ldv = _qkv_geometry(q, k, v, heads, hd, "attention")
    out, ldo = _out_like(q, out)
    check(lib.ymk_attention(DT[q.dtype], _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), ldo, B, Hq * Wq, Hk * Wk, heads, hd,
                            float(scale), _stream()), "attention")
    return out


@_timed("window_attention", _win_flops)
def window_attention(q, k, v, heads: int, hd: int, scale: float, win: int, shift: int = 0, pad_q=None, pad_k=None,
                     pad_v=None, out=None):
    """Attention inside win x win windows of the map padded (bottom / right) to a multiple of win; out-of-image tokens
    carry the fp32 vectors pad_q / pad_k / pad_v [heads*hd] (None = zeros) and take part as keys; with shift > 0 the
    padded grid is rolled by -shift in both axes before the partition and rolled back after (no mask).
    moa/heads.py:83-117, mot/experts.py:237-325."""
    B, H, W, Hk, Wk, ldq, ldk, ldv = _qkv_geometry(q, k, v, heads, hd, "window_attention")
    if (Hk, Wk) != (H, W):
        raise ValueError("window_attention: q, k, v live on one map")
    for t in (pad_q, pad_k, pad_v):
        if t is not None and (t.dtype != torch.float32 or t.numel() != heads * hd or not t.is_contiguous()):
            raise ValueError("window_attention: pad vectors are contiguous fp32 [heads*hd]")
    out, ldo = _out_like(q, out)
    check(lib.ymk_window_attention(DT[q.dtype], _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), ldo, B, H, W, heads, hd, float(scale),
                                   win, shift, _p(pad_q), _p(pad_k), _p(pad_v), _stream()), "window_attention")
    return out


@_timed("linear_attention", _lin_flops)
def linear_attention(q, k, v, rf, heads: int, hd: int, out=None):
    """ReLU random-feature attention of _GlobalAttnHead._linear_attn (moa/heads.py:318-352), fp32 math:
    phi(t) = min(relu(t rf^T / sqrt(nb)) + 1e-6, 1e4); out = clamp(phi(q) (phi(k)^T v), +-1e4) / max(phi(q) sum phi(k), 1e-6)."""
    B, H, W, Hk, Wk, ldq, ldk, ldv = _qkv_geometry(q, k, v, heads, hd, "linear_attention")
    if (Hk, Wk) != (H, W) or rf.dtype != torch.float32 or rf.shape[1] != hd or not rf.is_contiguous():
        raise ValueError("linear_attention: q, k, v on one map; rf a contiguous fp32 [nb, hd] matrix")
    nb = rf.shape[0]
    out, ldo = _out_like(q, out)
    nchunk = (H * W + 511) // 512    # csrc/mixattn.hip LINATTN_CHUNK: per (image, head) one reduced slot + one partial per 512-token chunk
    ws = torch.empty((B * heads * (nchunk + 1) * (nb * hd + nb),), dtype=torch.float32, device=q.device)
    check(lib.ymk_linear_attention(DT[q.dtype], _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(rf), nb, _p(out), ldo, B, H * W, heads, hd,
                                   _p(ws), _stream()), "linear_attention")
    return out


@_timed("deform_attention")
def deform_attention(v, off_logits, aw_logits, heads: int, hd: int, n_points: int, align_corners: bool, out=None):
    """_DeformableTransformerExpert._deform_attn (mot/experts.py:381-459): per token and head, locations =
    clamp(ref + 0.25 * tanh(off_logits), -1, 1) around the token's own normalised position, weights = softmax over the
    points of aw_logits, bilinear samples of v's head slice (zeros padding), weighted sum.  Coordinates fp32."""
    B, H, W, Cc, ldv = _nhwc(v)
    if off_logits.dtype != torch.float32 or aw_logits.dtype != torch.float32 or Cc != heads * hd:
        raise ValueError("deform_attention: fp32 offsets / weights, v [B,H,W,heads*hd]")
    ldoff, ldaw = _nhwc(off_logits)[4], _nhwc(aw_logits)[4]
    out, ldo = _out_like(v, out)
    check(lib.ymk_deform_attention(DT[v.dtype], _p(v), ldv, _p(off_logits), ldoff, _p(aw_logits), ldaw, _p(out), ldo, B, H, W, heads,
                                   hd, n_points, int(bool(align_corners)), _stream()), "deform_attention")
    return out


@_timed("token_router")
def token_softmax(logits, n: int, inv_temp: float, top_k: int = 0, out=None, bias=None, shape=None):
    """Per-token softmax over the first n channels of fp32 logits scaled by inv_temp; with 0 < top_k < n the top_k
    largest are kept and renormalised (sum clamped at 1e-6), the rest set to 0 (mot/router.py:243-295, moa/router.py:50-62).
    bias: fp32 [B, n] added to every token's logits of its image first (scene-aware residual, mot/router.py:224-240); logits=None with
    shape=(B, H, W): the image-level router, whose logits are the bias alone.
    Returns (weights fp32 [B,H,W,n], active int32 [B,n] = 1 where any token of the image gives expert e a nonzero weight)."""
    if logits is None:
        if bias is None or shape is None:
            raise ValueError("token_softmax: logits or (bias and shape)")
        B, H, W = shape
        ldl, dev = 0, bias.device
        if out is None:
            out = torch.empty((B, H, W, n), dtype=torch.float32, device=dev)
        ldw = _nhwc(out)[4]
    else:
        B, H, W, Cc, ldl = _nhwc(logits)
        if logits.dtype != torch.float32 or Cc < n:
            raise ValueError("token_softmax: fp32 logits with at least n channels")
        out, ldw = _out_like(logits, out, torch.float32, (B, H, W, n))
        dev = logits.device
    if bias is not None and not (bias.dtype == torch.float32 and bias.is_contiguous() and tuple(bias.shape) == (B, n)):
        raise ValueError("token_softmax: bias fp32 [B, n]")
    active = torch.zeros((B, n), dtype=torch.int32, device=dev)
    check(lib.ymk_token_softmax(_p(logits), ldl, _p(bias), _p(out), ldw, _p(active), B, H * W, n, float(inv_temp), int(top_k), _stream()),
          "token_softmax")
    return out, active


@_timed("pool")
def scene_bias(x, w1, b1, w2, b2, base=None):
    """Scene statistics of the routed map + the scene projector (mot/router.py:166-192, 224-240; include/ymk_mixture.h ymk_scene_bias).
    x [B,H,W,C]; w1 [hidden,3], b1 [hidden], w2 [E,hidden], b2 [E] fp32; base: None or fp32 [B,E] added to the bias.
    Returns (stats fp32 [B,3] = (high_frequency, heterogeneity, multi_scale), bias fp32 [B,E])."""
    B, H, W, Cc, ldx = _nhwc(x)
    hidden, E = w1.shape[0], w2.shape[0]
    dev = x.device
    cs = channel_stats(x, want_std=True)
    p4 = adaptive_avg_pool(x, min(4, H), min(4, W), out_dtype=torch.float32)
    p2 = adaptive_avg_pool(x, min(2, H), min(2, W), out_dtype=torch.float32)
    stats = torch.empty((B, 3), dtype=torch.float32, device=dev)
    bias = torch.empty((B, E), dtype=torch.float32, device=dev)
    nbytes = lib.ymk_scene_workspace_bytes(B, H)
    ws = torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=dev)
    check(lib.ymk_scene_bias(DT[x.dtype], _p(x), ldx, B, H, W, Cc, _p(cs), _p(p4), _p(p2), _p(w1), _p(b1), _p(w2), _p(b2), hidden, E,
                             _p(base), _p(stats), _p(bias), _p(ws), ws.numel(), _stream()), "scene_bias")
    return stats, bias


def moa_sparse_gate(weights, n: int, threshold: float):
    """MoA sparse inference (moa/block.py:194-234): which of the n head groups run for this batch, and their renormalised per-token
    gates.  weights fp32 [B,H,W,>=n] (token_softmax).  Returns (active: list of bool — ONE host sync, as in the reference, which reads
    the decision with bool(...) too —, blend fp32 [B,H,W,n] with the active groups' gates in its first columns, mass: mean gate sum of the
    skipped groups, the reference's `dropped_routing_mass` diagnostic)."""
    B, H, W, Cc, ldw = _nhwc(weights)
    if weights.dtype != torch.float32 or Cc < n:
        raise ValueError("moa_sparse_gate: fp32 gates with at least n channels")
    dev = weights.device
    stats = torch.zeros((2 * n,), dtype=torch.float64, device=dev)          # n sums (fp64) + n maxima (the bits of non-negative floats)
    blend = torch.empty((B, H, W, n), dtype=torch.float32, device=dev)
    active = torch.empty((n,), dtype=torch.int32, device=dev)
    if torch.cuda.is_available() and weights.is_cuda and torch.cuda.is_current_stream_capturing():
        # the decision is read on the host (as the reference does, moa/block.py:196-202): not expressible inside a captured HIP graph
        raise RuntimeError("MoABlock(sparse_inference=True) decides on the host which head groups run: it cannot be captured into a HIP graph "
                           "(bench.py / serving loops replay graphs) — construct the block with sparse_inference=False for graph replay, or run eagerly")
    check(lib.ymk_moa_sparse_gate(_p(weights), ldw, B * H * W, n, float(threshold), _p(stats), _p(blend), n, _p(active), _stream()),
          "moa_sparse_gate")
    act = [bool(v) for v in active.tolist()]
    sums = stats[:n].tolist()
    mass = sum(s_ for s_, a_ in zip(sums, act) if not a_) / float(B * H * W)
    return act, blend, mass


@_timed("token_router")
def gated_route_decide(g_logits, loc_logits, alpha: float, inv_temp: float, top_k: int, cplx_logit, clamp: int = 1):
    """Decision tail of the gated MoE (moe/gated.py:124-166, 455-492): logits = clamp(a*g + (1-a)*loc, +-30) with
    a = sigmoid(alpha); probs = softmax(logits * inv_temp) (clamp=1; clamp=2: the clamp AFTER the scaling, gated.py:972; clamp=0: a
    router's own plain softmax, gated.py:958 / routers.py:207); top-k, weights / (sum + 1e-6); complexity = clamp(mean_b
    sigmoid(cplx_logit[b]), 0.3, 1.5) (1.0 when non-finite) keeps round(c * top_k) in [1, top_k] ranked experts and
    renormalises (clamp 1e-6).  Inputs fp32 [B,1,1,E] / [B,1,1,1]; returns (w fp32 [B,1,1,top_k], idx int32 [B,top_k], probs,
    rows int32 [top_k*B] = idx transposed: the expert of image j*B + b in expert_conv's slot-major output)."""
    B, E = g_logits.shape[0], g_logits.shape[-1]
    for t in (g_logits, loc_logits, cplx_logit):
        if t.dtype != torch.float32 or t.shape[0] != B or t.shape[1:3] != (1, 1) or t.stride(3) != 1:
            raise ValueError("gated_route_decide: fp32 [B,1,1,*] inputs")
    dev = g_logits.device
    w = torch.empty((B, 1, 1, top_k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, top_k), dtype=torch.int32, device=dev)
    rows = torch.empty((top_k * B,), dtype=torch.int32, device=dev)
    probs = torch.empty((B, E), dtype=torch.float32, device=dev)
    check(lib.ymk_gated_route_decide(_p(g_logits), g_logits.stride(0), _p(loc_logits), loc_logits.stride(0), _p(cplx_logit),
                                     cplx_logit.stride(0), B, E, float(alpha), float(inv_temp), int(clamp), int(top_k), _p(w), _p(idx), _p(rows),
                                     _p(probs), _stream()), "gated_route_decide")
    return w, idx, probs, rows


@_timed("token_router")
def pooled_softmax_route(logits, E: int, inv_temp: float, top_k: int, threshold: float):
    """UltraEfficientRouter's decision (moe/routers.py:117-147) + the inference threshold on the routed weights (moe/utils.py:166-169):
    logits fp32 NHWC [B,h,w,>=E] of the router map -> (w fp32 [B,1,1,top_k] with weights <= threshold zeroed, idx int32 [B,top_k],
    pooled fp32 [B,E] = the per-pixel softmax averaged over the pixels, rows int32 [top_k*B])."""
    B, H, W, Cc, ldl = _nhwc(logits)
    if logits.dtype != torch.float32 or Cc < E:
        raise ValueError("pooled_softmax_route: fp32 logits with at least E channels")
    dev = logits.device
    w = torch.empty((B, 1, 1, top_k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, top_k), dtype=torch.int32, device=dev)
    rows = torch.empty((top_k * B,), dtype=torch.int32, device=dev)
    pooled = torch.empty((B, E), dtype=torch.float32, device=dev)
    check(lib.ymk_pooled_softmax_route(_p(logits), ldl, B, H * W, E, float(inv_temp), int(top_k), float(threshold), _p(w), _p(idx), _p(rows),
                                       _p(pooled), _stream()), "pooled_softmax_route")
    return w, idx, pooled, rows


def batch_scale(w, logit, lo: float, hi: float):
    """In place: w *= clamp(mean_b sigmoid(logit[b]), lo, hi) (1 when the mean is not finite) — the batch-level complexity scale of
    UltimateOptimizedMoE's routing weights (moe/modules.py:1662-1672).  w fp32 [B,1,1,K] dense, logit fp32 [B,1,1,>=1]."""
    _need_gpu(w)
    B, K = w.shape[0], w.shape[-1]
    if w.dtype != torch.float32 or logit.dtype != torch.float32 or not w.is_contiguous() or logit.shape[0] != B:
        raise ValueError("batch_scale: dense fp32 weights [B,1,1,K], fp32 logits [B,1,1,*]")
    check(lib.ymk_batch_scale(_p(w), B, K, _p(logit), logit.stride(0), float(lo), float(hi), _stream()), "batch_scale")
    return w


@_timed("expert_dw3")
def expert_dw3(x, w, dil, idx, out=None):
    """Per-image expert depthwise 3x3 with per-expert dilation, slot-major (include/ymk_mixture.h ymk_expert_dw3): x NHWC [B, H, W, C],
    w [E, 9, C] in x's dtype, dil int32 [E], idx int32 [B, K] -> [K * B, H, W, C] (DiversifiedExpertGroup.dw_layers, gated.py:2265-2278)."""
    B, H, W, Cc, ldx = _nhwc(x)
    E, K = w.shape[0], idx.shape[1]
    if idx.dtype != torch.int32 or tuple(idx.shape) != (B, K) or not idx.is_contiguous() or dil.dtype != torch.int32 or w.dtype != x.dtype:
        raise ValueError("expert_dw3: idx int32 [B, K] contiguous, dil int32 [E], w in the activation dtype")
    if out is None:
        out = torch.empty((K * B, H, W, Cc), dtype=x.dtype, device=x.device)
    check(lib.ymk_expert_dw3(DT[x.dtype], _p(x), ldx, _p(w), _p(dil), _p(idx), B, H, W, Cc, K, E, _p(out), _stream()), "expert_dw3")
    return out


@_timed("expert_conv")
def expert_conv(x, w_packed, k: int, idx, out=None):
    """Per-image expert convolution, slot-major: out[j*B + b] = conv_kxk(x[b], w_packed[idx[b, j]]) (no bias, no activation;
    out[j*B:(j+1)*B] is the batch of slot j, a contiguous NHWC tensor);
    w_packed [E][Cout][Kpad] in the compute dtype, idx int32 [B][K].  The selected slices of FusedExpertGroup's grouped
    3x3 (moe/gated.py:1058-1076; grouped weights expanded to dense rows at pack time) and the expert projections of
    SharedInvertedExpertGroup (moe/experts.py:235-269).  bf16 with Cin, Cout multiples of 64: TRUE sparse dispatch on the LDS-DMA
    tiled core (ymk_expert_conv_glds: tiles per (slot, image), the routed filter bank chosen on the device) — only the routed
    experts' MACs run (validated on MI355X, tests/test_gpu_next.py::test_expert_conv_glds_direct; YMK_DISABLE bit 512 switches
    it off).  Other shapes / fp32: all experts' rows as one ymk_conv2d (what the reference's fused convolution does) + a gather."""
    B, H, W, Cin, _ = _nhwc(x)
    E, Cout, Kp = w_packed.shape
    K = idx.shape[1]
    if idx.dtype != torch.int32 or tuple(idx.shape) != (B, K) or not idx.is_contiguous():
        raise ValueError("expert_conv: idx is a contiguous int32 [B, K] tensor")
    if out is None:
        out = torch.empty((K * B, H, W, Cout), dtype=x.dtype, device=x.device)
    if not out.is_contiguous():
        raise ValueError("expert_conv: dense output")
    if OPTIONS.expert_conv_glds and x.dtype in H16 and Cin % 64 == 0 and Cout % 64 == 0 and Kp == k * k * Cin:
        # true sparse dispatch on the LDS-DMA tiled core (include/ymk_next.h): only the routed filter banks run
        d = ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, B, H, W, Cin, Cout, k, 1, _nhwc(x)[4], Cout, 0, Kp, _lib.ACT_NONE)
        check(lib.ymk_expert_conv_glds(C.byref(d), _p(x), _p(w_packed), _p(idx), K, E, _p(out),
                                       1, _stream()), "expert_conv_glds")   # two LDS stages: the faster loop on every 3x3 shape measured
        return out
    zero_b = torch.zeros((E * Cout,), dtype=torch.float32, device=x.device)
    f_all = conv2d(x, w_packed.reshape(E * Cout, Kp), zero_b, k, 1, False)
    check(lib.ymk_expert_gather(DT[x.dtype], _p(f_all), _nhwc(f_all)[4], _p(idx), B, H * W, Cout, K, E, _p(out), _stream()),
          "expert_gather")
    return out


@_timed("layout")
def channel_shuffle_cat(parts, groups: int, out=None):
    """Channel concatenation followed by _channel_shuffle (moe/gated.py:1333-1338) in one pass:
    out[..., j * groups + i] = cat(parts)[..., i * (C / groups) + j]."""
    if len(parts) != 2 or parts[0].dtype != parts[1].dtype or parts[0].shape[:3] != parts[1].shape[:3]:
        raise ValueError("channel_shuffle_cat: two maps of one size and dtype")
    a, b = parts
    B, H, W, Ca, lda = _nhwc(a)
    Cb, ldb = b.shape[-1], _nhwc(b)[4]
    out, ldy = _out_like(a, out, None, (B, H, W, Ca + Cb))
    check(lib.ymk_channel_shuffle_cat(DT[a.dtype], _p(a), lda, Ca, _p(b), ldb, Cb, groups, _p(out), ldy, B * H * W, _stream()),
          "channel_shuffle_cat")
    return out


@_timed("layout")
def pixel_shuffle2(t, out=None):
    """Depth-to-space by 2: t [B,H,W,4C] (phase-major channel slices) -> [B,2H,2W,C]; with the 4C-channel 1x1 convolution in
    front it is Proto's ConvTranspose2d(C, C, 2, 2) (nn/modules/block.py:101-107)."""
    B, H, W, C4, ldt = _nhwc(t)
    Cc = C4 // 4
    out, ldo = _out_like(t, out, None, (B, 2 * H, 2 * W, Cc))
    check(lib.ymk_pixel_shuffle2(DT[t.dtype], _p(t), ldt, _p(out), ldo, B, H, W, Cc, _stream()), "pixel_shuffle2")
    return out


@_timed("layout")
def tokens_to_rows(x, y, a_off: int, row_off: int = 0):
    """y[b][row_off + c][a_off + p] = x[b][p][c] for an NHWC map x and an fp32 [B, rows, A] tensor y (mask coefficients of the
    Segment head in the reference's layout, nn/modules/head.py:341-349)."""
    B, H, W, Cc, ldx = _nhwc(x)
    if y.dtype != torch.float32 or not y.is_contiguous() or y.shape[0] != B:
        raise ValueError("tokens_to_rows: y is a contiguous fp32 [B, rows, A] tensor")
    check(lib.ymk_tokens_to_rows(DT[x.dtype], _p(x), ldx, _p(y), B, H * W, Cc, a_off, y.shape[2], row_off, y.shape[1], _stream()),
          "tokens_to_rows")
    return y
