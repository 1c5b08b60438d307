"""Module library of the MI355X-native YOLO-Master detection path.

Every class keeps the constructor signature, attribute names and parameter/buffer names of
its reference counterpart (cited per class), so a reference ``state_dict`` loads unchanged
and ``parse_model`` can build the reference's model YAMLs.  The arithmetic does NOT live
here: ``forward`` hands NHWC views to hand-written HIP kernels through the libymk C-ABI
(``yolo_master_amd.ops``).  There is no PyTorch/CPU implementation of the forward pass in
this package — on a CPU tensor (or without libymk.so) the ops raise.

Internal calling convention: ``m._run(x, out=None, ...)`` takes/returns NHWC tensors
``[B, H, W, C]`` (possibly channel slices of a wider buffer, which is how Concat/chunk are
made free); ``m.forward(x)`` is the public NCHW-logical wrapper (returns a channels-last
view, zero-copy).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from ..options import OPTIONS


class _Opt:
    """A module attribute that follows `options.OPTIONS.<name>` at CALL time (setting OPTIONS after import takes effect)
    unless the instance has been given its own value (`m.chunk_mb = 2.0` in a test)."""

    def __init__(self, name):
        self.name = name

    def __set_name__(self, owner, attr):
        self.slot = "_opt_" + attr

    def __get__(self, obj, typ=None):
        if obj is not None and self.slot in obj.__dict__:
            return obj.__dict__[self.slot]
        return getattr(OPTIONS, self.name)

    def __set__(self, obj, value):
        obj.__dict__[self.slot] = value
from ..errors import MoERouterError, ShapeMismatchError

__all__ = [
    "autopad", "Conv", "DWConv", "Concat", "Bottleneck", "C2f", "C3", "C3k", "C3k2", "AAttn", "ABlock", "A2C2f",
    "DFL", "Detect", "Proto", "Segment", "DynamicRoutingLayer", "DepthwiseSeparableConv", "EfficientExpertGroup", "ES_MOE",
    "set_compute_dtype", "LazyUpsample",
]


def autopad(k, p=None, d=1):
    """'same' padding (ultralytics/nn/modules/conv.py:30-36)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class YmkModule(nn.Module):
    """Base: compute dtype + packed-weight cache + NCHW<->NHWC public wrapper."""

    ymk_dtype = torch.float32

    def _cache(self) -> dict:
        c = self.__dict__.get("_ymk_pack")
        if c is None:
            c = {}
            self.__dict__["_ymk_pack"] = c
        return c

    def clear_pack(self):
        self.__dict__["_ymk_pack"] = {}

    def _packed(self, device):
        key = (self.ymk_dtype, str(device))
        c = self._cache()
        if key not in c:
            with torch.no_grad():
                c[key] = self._pack(self.ymk_dtype, device)
        return c[key]

    def _pack(self, dtype, device):  # pragma: no cover - overridden
        raise NotImplementedError

    # public, reference-compatible entry: NCHW-logical in, NCHW-logical (channels-last) out
    def forward(self, x):
        if self.training:
            raise RuntimeError(
                f"{type(self).__name__}: the ymk path implements eval-mode inference only "
                "(training runs on the reference implementation)"
            )
        y = self._run(to_nhwc(x, self.ymk_dtype))
        return y.permute(0, 3, 1, 2)


def to_nhwc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """NCHW-logical tensor (any memory format) -> NHWC tensor of the compute dtype."""
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got {tuple(x.shape)}")
    ops.require_gpu(x, "yolo_master_amd modules")
    xh = x.permute(0, 2, 3, 1)
    if xh.dtype != dtype:
        xh = xh.to(dtype)
    return xh if xh.is_contiguous() else xh.contiguous()


def set_compute_dtype(model: nn.Module, dtype: torch.dtype) -> nn.Module:
    """Select fp32, bf16 or fp16 activations/weights for every ymk module (parameters stay fp32 masters).  fp16 is the reference's
    reduced-precision mode (`half=True`, engine/predictor.py:174,415) and runs on libymk_f16.so."""
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError("ymk compute dtype must be torch.float32, torch.bfloat16 or torch.float16")
    if dtype is torch.float16 and not ops.HAS_F16:
        raise RuntimeError("libymk_f16.so not found: build it with `python -m yolo_master_amd.build`")
    for m in model.modules():
        if isinstance(m, YmkModule):
            m.ymk_dtype = dtype
            m.clear_pack()
    return model


def _is_silu(act) -> bool:
    if isinstance(act, nn.SiLU):
        return True
    if isinstance(act, nn.Identity):
        return False
    raise NotImplementedError(f"ymk Conv supports SiLU / Identity activations, got {act}")


def _bn_scale_shift(bn: nn.BatchNorm2d):
    s = bn.weight.div(torch.sqrt(bn.running_var + bn.eps))
    return s, bn.bias - bn.running_mean * s


class Conv(YmkModule):
    """Conv2d + BatchNorm2d + SiLU (ultralytics/nn/modules/conv.py:39-89)."""

    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()

    # -- packing ---------------------------------------------------------------------
    def _geometry(self):
        cv = self.conv
        k, s = cv.kernel_size[0], cv.stride[0]
        if cv.kernel_size[0] != cv.kernel_size[1] or cv.stride[0] != cv.stride[1] or cv.dilation != (1, 1):
            raise NotImplementedError("ymk Conv: square kernels, equal strides, dilation 1 only")
        if cv.padding != (k // 2, k // 2):
            raise NotImplementedError("ymk Conv: padding must be k//2 ('same')")
        dw = cv.groups > 1
        if dw and not (cv.groups == cv.in_channels == cv.out_channels and s == 1):
            raise NotImplementedError("ymk Conv: groups must be 1 or depthwise (stride 1)")
        if not dw and k not in (1, 3):
            raise NotImplementedError("ymk Conv: dense kernels are 1x1 or 3x3")
        return k, s, dw

    def _folded(self):
        w = self.conv.weight.detach().float()
        if hasattr(self, "bn"):
            bn = self.bn
            return ops.fold_bn(w, bn.weight.float(), bn.bias.float(), bn.running_mean.float(), bn.running_var.float(),
                               bn.eps, None if self.conv.bias is None else self.conv.bias.float())
        b = self.conv.bias.detach().float() if self.conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
        return w, b

    cout_perm = None  # optional output-channel permutation applied at pack time (AAttn.qkv)
    # optional zero padding of the packed weights (ABlock.mlp at the l/x scales: hidden width int(1.2*dim) is not a
    # multiple of the 16-byte channel vector; padded outputs are SiLU(0) = 0 and meet zero weight columns downstream)
    pad_cout_to = None
    pad_cin_to = None
    # set by the model builder on the convolution whose output IS the input of an ES-MoE layer: ask the kernel for the router's pooled sums
    pool_out = False

    def _pack(self, dtype, device):
        k, s, dw = self._geometry()
        w, b = self._folded()
        w, b = w.to(device), b.to(device)
        if self.cout_perm is not None:
            perm = torch.as_tensor(self.cout_perm, device=device)
            w, b = w[perm], b[perm]
        if self.pad_cout_to is not None and self.pad_cout_to > w.shape[0]:
            extra = self.pad_cout_to - w.shape[0]
            w = torch.cat([w, w.new_zeros((extra, *w.shape[1:]))], 0)
            b = torch.cat([b, b.new_zeros(extra)], 0)
        if self.pad_cin_to is not None and self.pad_cin_to > w.shape[1]:
            w = torch.cat([w, w.new_zeros((w.shape[0], self.pad_cin_to - w.shape[1], *w.shape[2:]))], 1)
        if dw:
            return {"dw": True, "w": ops.pack_dw_weight(w, dtype), "b": b.contiguous(), "k": k, "s": s}
        if self.conv.in_channels <= 4:  # stem: fp32 [Cout][k*k*Cin], reads the NCHW input directly
            wk = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
            return {"stem": True, "w": wk, "wt": wk.t().contiguous(), "b": b.contiguous(), "k": k, "s": s}
        return {"dw": False, "w": ops.pack_conv_weight(w, dtype), "b": b.contiguous(), "k": k, "s": s}

    # -- execution -------------------------------------------------------------------
    def _run(self, x, out=None, residual=None, out_dtype=None):
        pk = self._packed(x.device)
        act = _is_silu(self.act)
        if pk.get("stem"):
            raise RuntimeError("stem Conv (Cin<=4) consumes the NCHW network input: use _run_stem")
        if pk["dw"]:
            return ops.dwconv2d(x, pk["w"], pk["b"], pk["k"], act, out=out, residual=residual)
        return ops.conv2d(x, pk["w"], pk["b"], pk["k"], pk["s"], act, out=out, residual=residual, out_dtype=out_dtype, pool=self.pool_out)

    def _run_stem(self, x_nchw, out=None):
        pk = self._packed(x_nchw.device)
        return ops.conv2d_stem(x_nchw, pk["w"], pk["b"], pk["k"], pk["s"], _is_silu(self.act), self.ymk_dtype, out=out,
                               wt=pk["wt"])

    def forward(self, x):
        if self.training:
            raise RuntimeError("Conv: the ymk path implements eval-mode inference only")
        if self.conv.in_channels <= 4 and self.conv.groups == 1:
            ops.require_gpu(x, "yolo_master_amd modules")
            return self._run_stem(x).permute(0, 3, 1, 2)
        return self._run(to_nhwc(x, self.ymk_dtype)).permute(0, 3, 1, 2)

    forward_fuse = forward  # BN is always folded at pack time (torch_utils.py:315-349 semantics)


class DWConv(Conv):
    """Depth-wise convolution (ultralytics/nn/modules/conv.py:185-199)."""

    def __init__(self, c1, c2, k=1, s=1, d=1, act=True):
        super().__init__(c1, c2, k, s, g=math.gcd(c1, c2), d=d, act=act)


class LazyUpsample:
    """Marker returned by the graph walker for nn.Upsample: the consumer (Concat) materialises
    the 2x nearest upsample straight into its channel slice."""

    def __init__(self, src):
        self.src = src

    def materialise(self, out=None):
        return ops.upsample2x(self.src, out=out)


class VirtualCat:
    """Channel concatenation that is never materialised: the single consumer (the 1x1 `cv1` of the following
    C2f/C3k2) reads its K dimension from both sources (ymk_conv1x1_cat2)."""

    def __init__(self, parts):
        self.parts = parts  # [x1 or LazyUpsample(x1), x2]

    def materialise(self):
        return Concat(1)._run(self.parts)


class Concat(nn.Module):
    """Concatenate along channels (ultralytics/nn/modules/conv.py:616-641)."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def _run(self, xs):
        if self.d != 1:
            raise NotImplementedError("ymk Concat: channel dimension only")
        shapes = [(x.src.shape[0], 2 * x.src.shape[1], 2 * x.src.shape[2], x.src.shape[3], x.src.dtype, x.src.device)
                  if isinstance(x, LazyUpsample) else (*x.shape, x.dtype, x.device) for x in xs]
        B, H, W = shapes[0][:3]
        ctot = sum(s[3] for s in shapes)
        buf = ops.new_act(B, H, W, ctot, shapes[0][4], shapes[0][5])
        c0 = 0
        for x, s in zip(xs, shapes):
            sl = buf[..., c0:c0 + s[3]]
            if isinstance(x, LazyUpsample):
                x.materialise(out=sl)
            else:
                ops.copy_channels(x, sl)
            c0 += s[3]
        return buf

    def forward(self, x):
        dt = x[0].dtype
        y = self._run([to_nhwc(t, dt) for t in x])
        return y.permute(0, 3, 1, 2)


class Bottleneck(YmkModule):
    """Standard bottleneck (ultralytics/nn/modules/block.py:462-486)."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def _run(self, x, out=None):
        # (both 3x3s of a 64-channel block as one kernel was built, validated and measured 10-25 % slower than the pair on the LDS-DMA core:
        # tools/micro/parked/bneck.hip.txt, profiles/r04_negative_results.txt item 16)
        h = self.cv1._run(x)
        return self.cv2._run(h, out=out, residual=x if self.add else None)


class C2f(YmkModule):
    """CSP bottleneck with 2 convolutions (ultralytics/nn/modules/block.py:293-325)."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def _fusable(self, x):
        m0 = self.m[0]
        convs = (self.cv1, m0.cv1, m0.cv2, self.cv2)
        if not all(_is_silu(q.act) and q.conv.groups == 1 and q.conv.stride == (1, 1) for q in convs):
            return False
        if (self.cv1.conv.kernel_size, m0.cv1.conv.kernel_size, m0.cv2.conv.kernel_size, self.cv2.conv.kernel_size) != ((1, 1), (3, 3), (3, 3), (1, 1)):
            return False
        if m0.cv1.conv.out_channels * 2 != self.c:
            return False
        return ops.c3k2_fused_supported(x.dtype, self.cv1.conv.in_channels, self.cv2.conv.out_channels, self.c, 1, False, m0.add)

    def _run(self, x, out=None):
        # chunk(2)/cat are free: cv1 and every block write their slice of one buffer
        if isinstance(x, VirtualCat):
            p0, p1 = x.parts
            ref = p1
            B, H, W = ref.shape[0], ref.shape[1], ref.shape[2]
        else:
            ref = x
            B, H, W, _ = x.shape
        c, n = self.c, len(self.m)
        if torch.is_tensor(x) and n == 1 and type(self.m[0]) is Bottleneck and self._fusable(x):
            # the whole block as one kernel: cv1's output, the bottleneck's hidden map and its result stay in LDS (csrc/c3k2f.hip)
            m0 = self.m[0]
            pk = [q._packed(x.device) for q in (self.cv1, m0.cv1, m0.cv2, self.cv2)]
            return ops.c3k2_fused(x, *[(q["w"], q["b"]) for q in pk], out=out)
        cat = ops.new_act(B, H, W, (2 + n) * c, ref.dtype, ref.device)
        if isinstance(x, VirtualCat):
            pk = self.cv1._packed(ref.device)
            up = isinstance(p0, LazyUpsample)
            ops.conv1x1_cat2(p0.src if up else p0, up, p1, pk["w"], pk["b"], _is_silu(self.cv1.act), out=cat[..., : 2 * c])
        else:
            self.cv1._run(x, out=cat[..., : 2 * c])
        for i, m in enumerate(self.m):
            m._run(cat[..., (1 + i) * c:(2 + i) * c], out=cat[..., (2 + i) * c:(3 + i) * c])
        return self.cv2._run(cat, out=out)


class C3(YmkModule):
    """CSP bottleneck with 3 convolutions (ultralytics/nn/modules/block.py:327-351)."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=((1, 1), (3, 3)), e=1.0) for _ in range(n)))

    def _pack(self, dtype, device):
        """cv1 and cv2 are two 1x1 convolutions (+ BN + SiLU) of the same input: packed as ONE convolution with 2 c_ outputs — one launch
        and one read of x instead of two (the head's four C3k blocks of the S detector)."""
        (w1, b1), (w2, b2) = self.cv1._folded(), self.cv2._folded()
        w, b = torch.cat([w1, w2], 0).to(device), torch.cat([b1, b2], 0).to(device)
        return {"w": ops.pack_conv_weight(w, dtype), "b": b.contiguous()}

    def _pair_mergeable(self):
        a, b = self.cv1, self.cv2
        return (a.conv.kernel_size == b.conv.kernel_size == (1, 1) and a.conv.stride == b.conv.stride == (1, 1)
                and a.conv.groups == b.conv.groups == 1 and _is_silu(a.act) and _is_silu(b.act) and a.cout_perm is None and b.cout_perm is None)

    def _run(self, x, out=None):
        B, H, W, _ = x.shape
        c_ = self.cv1.conv.out_channels
        cat = ops.new_act(B, H, W, 2 * c_, x.dtype, x.device)
        n = len(self.m)
        # n >= 2 only: with ONE Bottleneck its input / residual (cat[..., :c_]) would also be its output — ymk_conv2d takes y and the
        # residual as __restrict__ pointers and does not promise that a core reads each residual element in the thread that writes it
        if n > 1 and self._pair_mergeable():
            self.ymk_dtype = self.cv1.ymk_dtype
            pk = self._packed(x.device)
            ops.conv2d(x, pk["w"], pk["b"], 1, 1, True, out=cat)     # cat = [cv1(x) | cv2(x)]; the last block overwrites the first half
            h = cat[..., :c_]
            for j, blk in enumerate(self.m):
                h = blk._run(h, out=cat[..., :c_] if j == n - 1 else None)
            return self.cv3._run(cat, out=out)
        h = self.cv1._run(x, out=cat[..., :c_] if n == 0 else None)
        for j, blk in enumerate(self.m):
            h = blk._run(h, out=cat[..., :c_] if j == n - 1 else None)
        self.cv2._run(x, out=cat[..., c_:])
        return self.cv3._run(cat, out=out)


class C3k(C3):
    """C3 with k x k bottlenecks (ultralytics/nn/modules/block.py:1114-1132)."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5, k=3):
        super().__init__(c1, c2, n, shortcut, g, e)
        c_ = int(c2 * e)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=(k, k), e=1.0) for _ in range(n)))


class C3k2(C2f):
    """C2f whose blocks are Bottleneck or C3k (ultralytics/nn/modules/block.py:1074-1111)."""

    def __init__(self, c1, c2, n=1, c3k=False, e=0.5, attn=False, g=1, shortcut=True):
        super().__init__(c1, c2, n, shortcut, g, e)
        if attn:
            raise NotImplementedError("ymk C3k2: attn=True (PSABlock) is not on the master det path")
        self.m = nn.ModuleList(
            C3k(self.c, self.c, 2, shortcut, g) if c3k else Bottleneck(self.c, self.c, shortcut, g) for _ in range(n)
        )


class AAttn(YmkModule):
    """Area attention (ultralytics/nn/modules/block.py:1646-1732)."""

    def __init__(self, dim, num_heads, area=1):
        super().__init__()
        self.area = area
        self.num_heads = num_heads
        self.head_dim = head_dim = dim // num_heads
        self.all_head_dim = all_head_dim = head_dim * self.num_heads
        self.qkv = Conv(dim, all_head_dim * 3, 1, act=False)
        self.proj = Conv(all_head_dim, dim, 1, act=False)
        self.pe = Conv(all_head_dim, all_head_dim, 7, 1, 3, g=all_head_dim, act=False)
        if head_dim != 32:
            raise NotImplementedError("ymk AAttn kernel is specialised for head_dim == 32")
        # re-order qkv output channels from per-head [q|k|v] to [Q(all heads) | K | V] so that q, k, v
        # of the 1x1 conv output are contiguous channel ranges (free: applied to the packed weights)
        d, h = head_dim, num_heads
        self.qkv.cout_perm = [hh * 3 * d + part * d + dd for part in range(3) for hh in range(h) for dd in range(d)]

    def _pre_proj(self, x):
        """attn(x) + pe(v): the projection's input (block.py:1696-1731)."""
        B, H, W, _ = x.shape
        if (H * W) % self.area:
            raise ValueError(f"AAttn: {H}x{W} tokens not divisible by area={self.area}")
        c = self.all_head_dim
        if (x.shape[-1] == c and self.qkv.conv.kernel_size == (1, 1) and not _is_silu(self.qkv.act)
                and ops.area_attn_qkv_supported(x.dtype, c, self.num_heads, H * W, self.area)):
            # the projection inside the attention kernel: K / V^T never leave the CU, only v (pe's input) and the result are stored
            pk = self.qkv._packed(x.device)
            att, v = ops.area_attn_qkv(x, pk["w"], pk["b"], self.num_heads, self.area)
            return self.pe._run(v, residual=att)
        qkv = self.qkv._run(x)
        att = ops.area_attn(qkv, self.num_heads, self.area)
        return self.pe._run(qkv[..., 2 * c:], residual=att)  # x + pe(v)

    def _run(self, x, out=None, residual=None):
        """Returns residual + proj(attn(x) + pe(v)) (residual = the ABlock skip input)."""
        return self.proj._run(self._pre_proj(x), out=out, residual=residual)


class ABlock(YmkModule):
    """Area-attention block (ultralytics/nn/modules/block.py:1735-1797)."""

    def __init__(self, dim, num_heads, mlp_ratio=1.2, area=1):
        super().__init__()
        self.attn = AAttn(dim, num_heads=num_heads, area=area)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(Conv(dim, mlp_hidden_dim, 1), Conv(mlp_hidden_dim, dim, 1, act=False))
        hp = (mlp_hidden_dim + 7) // 8 * 8
        # ... and, where it costs at most 1/8 more arithmetic, to a multiple of 64: both 1x1 convolutions then qualify for the LDS-DMA core
        # (int(1.2 * 256) = 307 -> 320 at the L scale: config 5 309.4 -> 311 images/s on one box; the generic implicit-GEMM kernel ran them before)
        hp64 = (mlp_hidden_dim + 63) // 64 * 64
        if dim % 64 == 0 and hp64 - mlp_hidden_dim <= mlp_hidden_dim // 8:
            hp = hp64
        if hp != mlp_hidden_dim:
            self.mlp[0].pad_cout_to = hp
            self.mlp[1].pad_cin_to = hp
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Conv2d):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def _run(self, x, out=None):
        m0, m1 = self.mlp[0], self.mlp[1]
        x1 = self.attn._run(x, residual=x)              # x + attn(x)
        if ops.mlp_fused_supported(x1.dtype, x1.shape[-1], m0.conv.out_channels) and _is_silu(m0.act) and not _is_silu(m1.act):
            p0, p1 = m0._packed(x1.device), m1._packed(x1.device)   # x + mlp(x) as one kernel: the hidden tensor never leaves the CU
            return ops.mlp_fused(x1, p0["w"], p0["b"], p1["w"], p1["b"], out=out)
        h = m0._run(x1)
        return m1._run(h, out=out, residual=x1)  # x + mlp(x)


class A2C2f(YmkModule):
    """Area-attention C2f (ultralytics/nn/modules/block.py:1800-1879)."""

    def __init__(self, c1, c2, n=1, a2=True, area=1, residual=False, mlp_ratio=2.0, e=0.5, g=1, shortcut=True):
        super().__init__()
        c_ = int(c2 * e)
        assert c_ % 32 == 0, "Dimension of ABlock must be a multiple of 32."
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv((1 + n) * c_, c2, 1)
        self.gamma = nn.Parameter(0.01 * torch.ones(c2), requires_grad=True) if a2 and residual else None
        self.m = nn.ModuleList(
            nn.Sequential(*(ABlock(c_, c_ // 32, mlp_ratio, area) for _ in range(2))) if a2 else C3k(c_, c_, 2, shortcut, g)
            for _ in range(n)
        )

    def _run(self, x, out=None):
        B, H, W, _ = x.shape
        c_ = self.cv1.conv.out_channels
        n = len(self.m)
        cat = ops.new_act(B, H, W, (1 + n) * c_, x.dtype, x.device)
        self.cv1._run(x, out=cat[..., :c_])
        for i, m in enumerate(self.m):
            src, dst = cat[..., i * c_:(i + 1) * c_], cat[..., (i + 1) * c_:(i + 2) * c_]
            if isinstance(m, nn.Sequential):
                h = src
                for j, blk in enumerate(m):
                    h = blk._run(h, out=dst if j == len(m) - 1 else None)
            else:
                m._run(src, out=dst)
        if self.gamma is None:
            return self.cv2._run(cat, out=out)
        # l/x scales: x + gamma * cv2(...) (block.py:1877-1879)
        y = self.cv2._run(cat)
        return ops.scale_residual(y, self.gamma.detach().float().contiguous(), x, out=out)


class DFL(nn.Module):
    """Distribution-focal-loss integral (ultralytics/nn/modules/block.py:63-84); the decode
    kernel evaluates it in closed form, this module only carries the fixed arange weights."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        x = torch.arange(c1, dtype=torch.float)
        self.conv.weight.data[:] = nn.Parameter(x.view(1, c1, 1, 1))
        self.c1 = c1


class _PlainConv:
    """Packing helper for the bare nn.Conv2d 1x1 (+bias) that ends each Detect branch."""

    @staticmethod
    def pack(conv: nn.Conv2d, dtype, device):
        """Output channels are zero-padded to a multiple of 4 (ymk_conv2d's 16-byte fp32 store granularity): `nc` / `nm`
        are free ctor arguments of the reference (nc=1, 2, 3 ... datasets); callers slice the padded channels off."""
        if conv.kernel_size != (1, 1) or conv.groups != 1:
            raise NotImplementedError("Detect tail conv must be 1x1")
        w = conv.weight.detach().float().to(device)
        b = conv.bias.detach().float().to(device) if conv.bias is not None else torch.zeros(w.shape[0], device=device)
        co = w.shape[0]
        if co % 4:
            w = torch.cat([w, w.new_zeros((4 - co % 4, *w.shape[1:]))], 0)
            b = torch.cat([b, b.new_zeros(4 - co % 4)])
        return ops.pack_conv_weight(w, dtype), b.contiguous()


class DetectPreds(dict):
    """The dict Detect returns beside y in eval mode (nn/modules/head.py:157-171): "boxes" [B, 4*reg_max, A] and "scores"
    [B, nc, A] raw logits, "feats" = the head's input maps (NCHW-logical).  Nothing on the inference path reads them
    (predict()/val() consume y), so the three reference keys are built on first access from the per-level NHWC logits
    ("raw", what the decode consumed) instead of being concatenated every step: layout changes only, no arithmetic.  With the fused
    decode (Detect.fuse_decode, keep_raw False) "raw" itself is lazy: Detect.raw_logits recomputes it from "feats"."""

    def __init__(self, raw, feats, reg_max, nc, raw_fn=None):
        super().__init__()
        self._lazy = {"boxes": lambda: self._cat(0, 4 * reg_max), "scores": lambda: self._cat(1, nc),
                      "feats": lambda: [f.permute(0, 3, 1, 2) for f in feats]}
        if raw_fn is not None and any(r is None for r in raw):
            # fused decode (Detect._level): the logits stayed on chip; recomputed from the head's input maps on first access
            self._lazy["raw"] = raw_fn
        else:
            self["raw"] = raw

    def _cat(self, which, width):
        lv = [r[which] for r in self["raw"]]
        return torch.cat([t.reshape(t.shape[0], -1, width) for t in lv], 1).permute(0, 2, 1)

    def __missing__(self, key):
        if key not in self._lazy:
            raise KeyError(key)
        self[key] = self._lazy[key]()
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(dict.keys(self)) + [k for k in self._lazy if not dict.__contains__(self, k)]


class Detect(YmkModule):
    """Detection head (ultralytics/nn/modules/head.py:37-258): box/cls branches + DFL decode."""

    dynamic = False
    export = False
    format = None
    max_det = 300
    agnostic_nms = False
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)
    legacy = False
    xyxy = False

    def __init__(self, nc=80, reg_max=16, end2end=False, ch=()):
        super().__init__()
        if end2end:
            raise NotImplementedError("ymk Detect: end2end (one2one) heads are not on the master det path")
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = reg_max
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch
        )
        self.cv3 = (
            nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
            if self.legacy
            else nn.ModuleList(
                nn.Sequential(
                    nn.Sequential(DWConv(x, x, 3), Conv(x, c3, 1)),
                    nn.Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)),
                    nn.Conv2d(c3, self.nc, 1),
                )
                for x in ch
            )
        )
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()

    end2end = False

    def bias_init(self):
        """Detect bias init (head.py:196-211); requires self.stride."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 2.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / float(s)) ** 2)

    def _pack(self, dtype, device):
        return {"box": [_PlainConv.pack(s[-1], dtype, device) for s in self.cv2],
                "cls": [_PlainConv.pack(s[-1], dtype, device) for s in self.cv3]}

    # levels 1.. on side HIP streams (fork / join, also valid under graph capture): measured slower in round 1 (10.8 vs 10.1 ms/step,
    # contention) -> off; options.OPTIONS.detect_level_streams (YMK_ENABLE bit 16) switches it on for A/B runs
    level_streams = _Opt("detect_level_streams")

    def _side_streams(self, device, n, main=None):
        """Side streams of the walk that runs on stream `main` (two concurrent walks of one model — bench.py --split — must not share them)."""
        st = self.__dict__.setdefault("_ymk_streams", {})
        key = (str(device), None if main is None else main.cuda_stream)
        if key not in st or len(st[key]) < n:
            st[key] = st.get(key, []) + [torch.cuda.Stream(device=device) for _ in range(n - len(st.get(key, [])))]
        return st[key][:n]

    def _branch(self, seq, x):
        for m in list(seq)[:-1]:
            if isinstance(m, nn.Sequential):
                for mm in m:
                    x = mm._run(x)
            else:
                x = m._run(x)
        return x

    def _cls_fusable(self, i, f):
        seq = self.cv3[i]
        if self.legacy or len(seq) != 3 or not all(isinstance(seq[j], nn.Sequential) and len(seq[j]) == 2 for j in (0, 1)):
            return False
        d1, p1, d2, p2 = seq[0][0], seq[0][1], seq[1][0], seq[1][1]
        ok = all(isinstance(m, DWConv) and m.conv.kernel_size == (3, 3) and m.conv.stride == (1, 1) and m.conv.groups == m.conv.in_channels
                 and _is_silu(m.act) for m in (d1, d2))
        ok = ok and all(type(m) is Conv and m.conv.kernel_size == (1, 1) and m.conv.groups == 1 and _is_silu(m.act) for m in (p1, p2))
        return ok and ops.detect_cls_fused_supported(f.dtype, d1.conv.in_channels, p1.conv.out_channels, self.nc) and \
            p2.conv.out_channels == p1.conv.out_channels

    # ---- the head as per-level chains -------------------------------------------------------------------------------------------
    # A level's chain (box branch, class branch, DFL decode into its anchor range of y) depends on ONE input map only.  The graph
    # walk (nn/tasks.py) therefore starts a level on a side HIP stream as soon as its map exists — P3 is ready six layers before the
    # head — so that the level's large kernels overlap the latency-bound 40^2 / 20^2 launches of the rest of the neck (fork / join by
    # events: parallel branches of the captured graph).  `begin` allocates y, `start_level` forks, `finish` runs what is left and joins.
    # Measured (round 3, profiles/r03_negative_results.txt): +0.3 % on the one-stream step (6.049 -> 6.032 ms) — the levels' kernels and the
    # neck's do not overlap enough to matter — and nothing on top of bench.py's two concurrent sub-batches: OFF; options.OPTIONS.detect_early_levels (YMK_ENABLE bit 32) for A/B runs.
    early_levels = _Opt("detect_early_levels")

    def begin(self, B, level_hw, device):
        """level_hw: [(H_l, W_l)] of every pyramid level.  Returns the run state (y, anchor offsets, raw slots)."""
        if self.reg_max <= 1:
            raise NotImplementedError("ymk Detect: reg_max must be > 1 (DFL)")
        offs, a_off = [], 0
        for h, w in level_hw:
            offs.append(a_off)
            a_off += h * w
        with torch.inference_mode(False):   # a normal tensor even under inference_mode: it keeps a version counter (see finish)
            y = torch.empty((B, 4 + self.nc, a_off), dtype=torch.float32, device=device)
        # every anchor's best class score and class, written by the decode kernel next to y: the single-label candidate filter of
        # non_max_suppression reads these 8 bytes per anchor instead of the nc class rows (ops.nms_batched looks for `y.best`)
        y.best = (torch.empty((B, a_off), dtype=torch.float32, device=device), torch.empty((B, a_off), dtype=torch.int32, device=device))
        return {"y": y, "offs": offs, "raw": [None] * len(level_hw), "hw": list(level_hw), "side": [], "main": None}

    # Fused decode (round 4): the last 1x1 of the box branch + DFL + dist2bbox and the class branch's sigmoid write y directly
    # (ops.detect_box_tail, ops.detect_cls_fused(y=...)): no detect_decode launch, and with keep_raw False the fp32 logits — 0.6 KB per
    # anchor written and read back — never reach HBM.  The reference's eval-mode `preds` (head.py:157-171) stays available: nothing
    # on the predict / val path reads it, so DetectPreds recomputes the logits on first access (raw_logits) from the head's inputs.
    # options.OPTIONS.fused_decode off (YMK_DISABLE bit 4194304): the unfused path; .detect_keep_raw (YMK_ENABLE bit 256): fused, logits materialised too.
    fuse_decode = True          # (ops.detect_box_tail_supported consults OPTIONS.fused_decode)
    keep_raw = _Opt("detect_keep_raw")

    def _cls_weights(self, i, device):
        s0, s1 = self.cv3[i][0], self.cv3[i][1]
        q = [m._packed(device) for m in (s0[0], s0[1], s1[0], s1[1])]
        return [(d["w"], d["b"]) for d in q]

    def _level(self, st, i, f, raw_only=False):
        pk = self._packed(f.device)
        if tuple(f.shape[1:3]) != tuple(st["hw"][i]):
            raise ValueError(f"Detect level {i}: map {tuple(f.shape[1:3])}, expected {st['hw'][i]}")
        hb = self._branch(self.cv2[i], f)
        cls_fused = self._cls_fusable(i, f)
        # (the fused pair's entry points also want 16-byte aligned bases and pixel strides that are multiples of 8 elements: true of every
        # buffer the walk hands over, checked here so that a foreign view takes the unfused path instead of failing with YMK_E_BADARG)
        ok16 = all(t.data_ptr() % 16 == 0 and (t.shape[2] == 1 or t.stride(2) % 8 == 0) for t in (hb, f))
        if not raw_only and self.fuse_decode and cls_fused and ok16 and ops.detect_box_tail_supported(hb.dtype, hb.shape[-1], self.reg_max, self.nc):
            y, keep = st["y"], self.keep_raw
            box = ops.detect_box_tail(hb, pk["box"][i][0], pk["box"][i][1], y, float(self.stride[i]), st["offs"][i], self.reg_max, raw=keep)
            cls = ops.detect_cls_fused(f, *self._cls_weights(i, f.device), pk["cls"][i], y=y, nc=self.nc, a_off=st["offs"][i], raw=keep)
            st["raw"][i] = (box, cls[..., : self.nc]) if keep else None
            return
        box = ops.conv2d(hb, pk["box"][i][0], pk["box"][i][1], 1, 1, False, out_dtype=torch.float32)
        if cls_fused:
            # the class branch of the level as one kernel: its four [B, H, W, 128] intermediates stay in LDS (csrc/detcls.hip)
            cls = ops.detect_cls_fused(f, *self._cls_weights(i, f.device), pk["cls"][i])[..., : self.nc]
        else:
            hc = self._branch(self.cv3[i], f)
            cls = ops.conv2d(hc, pk["cls"][i][0], pk["cls"][i][1], 1, 1, False, out_dtype=torch.float32)[..., : self.nc]
        if not raw_only:
            ops.detect_decode(box, cls, st["y"], float(self.stride[i]), st["offs"][i], self.reg_max, best=st["y"].best)
        st["raw"][i] = (box, cls)

    def raw_logits(self, feats):
        """Per-level (box logits [B,H,W,4*reg_max], class logits [B,H,W,nc]) fp32 of the head's NHWC input maps: what `preds["raw"]`
        holds.  With the fused decode they are not materialised by the forward pass; this recomputes them (same kernels, no decode)."""
        st = {"raw": [None] * len(feats), "hw": [tuple(f.shape[1:3]) for f in feats]}
        for i, f in enumerate(feats):
            self._level(st, i, f, raw_only=True)
        return st["raw"]

    def start_level(self, st, i, f):
        """Fork: level i on a side stream, ordered after everything enqueued so far on the current stream."""
        main = torch.cuda.current_stream()
        side = self._side_streams(f.device, len(st["side"]) + 1, main)[len(st["side"])]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._level(st, i, f)
        st["side"].append((side, i))
        st["main"] = main

    def finish(self, st, feats):
        """Run the levels that were not started early on the current stream, then join the side streams."""
        started = {i for _, i in st["side"]}
        for i, f in enumerate(feats):
            if i not in started:
                self._level(st, i, f)
        y = st["y"]
        # valid for this y as it stands now: ops.nms_batched compares the version counter, so an in-place edit by the caller drops them
        y.best = (y.best[0], y.best[1], -1 if y.is_inference() else y._version)
        if not st["side"]:
            return y, st["raw"]
        main = torch.cuda.current_stream()
        for side, i in st["side"]:
            main.wait_stream(side)
            for tns in st["raw"][i] or ():  # tensors created on a side stream are consumed on the main one
                tns.record_stream(main)
            for tns in (y, y.best[0], y.best[1]):
                tns.record_stream(side)
        return y, st["raw"]

    def _run(self, feats):
        """feats: list of NHWC maps.  Returns (y [B, 4+nc, A] fp32, raw) with raw = per-level
        (box_logits [B,H,W,4*reg_max], cls_logits [B,H,W,nc]) fp32 NHWC tensors."""
        st = self.begin(feats[0].shape[0], [tuple(f.shape[1:3]) for f in feats], feats[0].device)
        if self.level_streams and len(feats) > 1:
            # the pyramid levels are independent chains of small launches: run levels 1.. on side HIP streams so
            # they overlap the large level-0 kernels (fork/join with events; also valid under hipGraph capture)
            for i in range(1, len(feats)):
                self.start_level(st, i, feats[i])
        return self.finish(st, feats)

    def forward(self, x):
        if self.training:
            raise RuntimeError("Detect: the ymk path implements eval-mode inference only")
        feats = [to_nhwc(t, self.ymk_dtype) for t in x]
        y, raw = self._run(feats)
        if self.export:
            return y
        # the reference's eval-mode dict (head.py:157-171), lazily: with the fused decode the logits were never materialised, and nothing
        # on the predict / val path reads them — "boxes" / "scores" / "raw" re-run the head's branches on first access only
        preds = DetectPreds(raw, feats, self.reg_max, self.nc, raw_fn=lambda: self.raw_logits(feats))
        dict.__setitem__(preds, "feats", list(x))   # the maps as they were handed in
        return y, preds


class Proto(YmkModule):
    """Mask prototypes (ultralytics/nn/modules/block.py:88-107): Conv3x3 -> ConvTranspose2d(2, 2, bias) -> Conv3x3 -> Conv1x1.
    The transposed convolution runs as a 1x1 convolution to 4*c_ channels (one slice per output phase) + depth-to-space."""

    def __init__(self, c1, c_=256, c2=32):
        super().__init__()
        self.cv1 = Conv(c1, c_, k=3)
        self.upsample = nn.ConvTranspose2d(c_, c_, 2, 2, 0, bias=True)
        self.cv2 = Conv(c_, c_, k=3)
        self.cv3 = Conv(c_, c2)

    def _pack(self, dtype, device):
        w = self.upsample.weight.detach().float().to(device)           # [Cin, Cout, 2, 2]
        cin, cout = w.shape[:2]
        wr = w.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1)         # row (dy*2+dx)*Cout + co
        b = self.upsample.bias.detach().float().to(device).repeat(4)
        return {"up": (ops.pack_conv_weight(wr, dtype), b.contiguous())}

    def _run(self, x, out=None):
        pk = self._packed(x.device)
        t = ops.conv2d(self.cv1._run(x), *pk["up"], 1, 1, False)
        return self.cv3._run(self.cv2._run(ops.pixel_shuffle2(t)), out=out)


class Segment(Detect):
    """Segmentation head (ultralytics/nn/modules/head.py:265-349): Detect + mask-coefficient branches + prototypes.
    Device path: `_run` returns Detect's (y [B, 4+nc, A], raw) and leaves the mask coefficients fp32 [B, nm, A] and the
    prototypes NHWC [B, 2H0, 2W0, nm] in `last_mc` / `last_proto` (NMS consumes y; the kept anchors' coefficients are
    gathered by index afterwards).  `forward` returns the reference's eval structure ((cat(y, mc), proto), preds)."""

    def __init__(self, nc=80, nm=32, npr=256, reg_max=16, end2end=False, ch=()):
        super().__init__(nc, reg_max, end2end, ch)
        self.nm, self.npr = nm, npr
        self.proto = Proto(ch[0], self.npr, self.nm)
        c4 = max(ch[0] // 4, self.nm)
        self.cv4 = nn.ModuleList(nn.Sequential(Conv(x, c4, 3), Conv(c4, c4, 3), nn.Conv2d(c4, self.nm, 1)) for x in ch)

    def _pack(self, dtype, device):
        pk = super()._pack(dtype, device)
        pk["mask"] = [_PlainConv.pack(s[-1], dtype, device) for s in self.cv4]
        return pk

    def _run(self, feats):
        y, raw = super()._run(feats)
        pk = self._packed(feats[0].device)
        B, A = y.shape[0], y.shape[2]
        mc = torch.empty((B, self.nm, A), dtype=torch.float32, device=y.device)
        a_off = 0
        for i, f in enumerate(feats):
            h = self.cv4[i][1]._run(self.cv4[i][0]._run(f))
            c = ops.conv2d(h, pk["mask"][i][0], pk["mask"][i][1], 1, 1, False, out_dtype=torch.float32)[..., : self.nm]
            ops.tokens_to_rows(c, mc, a_off)
            a_off += f.shape[1] * f.shape[2]
        self.last_mc, self.last_proto = mc, self.proto._run(feats[0])
        return y, raw

    def forward(self, x):
        y, preds = super().forward(x)
        proto = ops.nhwc_to_nchw_f32(self.last_proto)
        preds["mask_coefficient"], preds["proto"] = self.last_mc, proto
        return (torch.cat([y, self.last_mc], 1), proto), preds   # the concatenation is API compatibility only


# ------------------------------------------------------------------------------ ES-MoE
class DynamicRoutingLayer(nn.Module):
    """Router parameters (ultralytics/nn/modules/moe/routers.py:429-457); evaluated by ymk_esmoe_route."""

    def __init__(self, in_channels, num_experts=3, reduction=8, top_k=None):
        super().__init__()
        if num_experts < 1:
            raise ValueError(f"num_experts must be positive, got {num_experts}")
        if reduction < 1:
            raise ValueError(f"reduction must be positive, got {reduction}")
        if top_k is not None and not 1 <= top_k <= num_experts:
            raise ValueError(f"top_k must be in [1, {num_experts}], got {top_k}")
        reduced_channels = max(in_channels // reduction, 8)
        self.in_channels = in_channels
        self.num_experts = num_experts
        self.top_k = min(top_k, num_experts) if top_k is not None else num_experts
        self.use_top_k = top_k is not None
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.routing_network = nn.Sequential(
            nn.Conv2d(in_channels, reduced_channels, kernel_size=1),
            nn.SiLU(inplace=False),
            nn.Conv2d(reduced_channels, num_experts, kernel_size=1),
        )
        self.last_routing_diagnostics = {}


class DepthwiseSeparableConv(nn.Module):
    """Expert body parameters (ultralytics/nn/modules/moe/experts.py:280-296)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1):
        super().__init__()
        padding = (kernel_size - 1) // 2
        self.depthwise = nn.Conv2d(in_channels, in_channels, kernel_size, stride=stride, padding=padding,
                                   groups=in_channels, bias=False)
        self.pointwise = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = nn.SiLU(inplace=True)


class EfficientExpertGroup(nn.Module):
    """ultralytics/nn/modules/moe/experts.py:299-311."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1):
        super().__init__()
        self.conv = DepthwiseSeparableConv(in_channels, out_channels, kernel_size, stride)


class ES_MOE(YmkModule):
    """Sample-routed conv MoE (ultralytics/nn/modules/moe/modules.py:410-779).

    Eval forward = router (GAP -> 2 x 1x1 -> softmax -> top-k -> threshold/renorm -> CSR), depthwise
    stage over the CSR pairs, pointwise grouped GEMM with fused BN/SiLU/gate/accumulate and the
    trailing BN+SiLU; all in libymk.  Non-finite router input/logits raise ``MoERouterError`` when the
    batch's device flag word is checked (``check_flags``), not through a per-layer host sync.
    """

    # (The expert body as ONE wave-specialised kernel per layer — halo staged once for both experts, stencil waves -> LDS tile -> matrix
    # waves — was built in round 4, is bit-identical to the two-kernel form and 1.65x slower: tools/micro/parked/esfused.hip.txt,
    # profiles/r04_esfused_ablation.txt.  It left the library and the ABI in round 5.)

    def __init__(self, in_channels, out_channels=None, num_experts=4, reduction=8, top_k=2, use_sparse_inference=True,
                 dynamic_threshold=0.4, max_kernel_size=15, expert_kernel_sizes=None):
        super().__init__()
        if in_channels < 1 or (out_channels is not None and out_channels < 1):
            raise ValueError("in_channels and out_channels must be positive")
        if num_experts < 1:
            raise ValueError(f"num_experts must be positive, got {num_experts}")
        if reduction < 1:
            raise ValueError(f"reduction must be positive, got {reduction}")
        if top_k is not None and not 1 <= top_k <= num_experts:
            raise ValueError(f"top_k must be in [1, {num_experts}], got {top_k}")
        if not 0.0 <= dynamic_threshold <= 1.0:
            raise ValueError(f"dynamic_threshold must be in [0, 1], got {dynamic_threshold}")
        if max_kernel_size < 3:
            raise ValueError(f"max_kernel_size must be at least 3, got {max_kernel_size}")
        max_kernel_size = int(max_kernel_size)
        if max_kernel_size % 2 == 0:
            max_kernel_size -= 1
        if out_channels is None:
            out_channels = in_channels
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_experts = num_experts
        self.reduction = reduction
        self.top_k = min(top_k, num_experts) if top_k is not None else num_experts
        self.use_top_k = top_k is not None
        self.use_sparse_inference = use_sparse_inference
        self.dynamic_threshold = dynamic_threshold
        self.max_kernel_size = max_kernel_size
        self.routing = DynamicRoutingLayer(in_channels, num_experts, reduction, top_k)
        if expert_kernel_sizes is not None:
            if len(expert_kernel_sizes) != num_experts:
                raise ValueError(f"expert_kernel_sizes must have {num_experts} entries, got {len(expert_kernel_sizes)}")
            ks = []
            for k in expert_kernel_sizes:
                k = int(k)
                if k % 2 == 0:
                    k -= 1
                ks.append(min(k, max_kernel_size))
        else:
            default_kernel_sizes = [3, 5, 7]
            if num_experts <= len(default_kernel_sizes):
                ks = [min(k, max_kernel_size) for k in default_kernel_sizes[:num_experts]]
            else:
                ks = [min(3 + 2 * i, max_kernel_size) for i in range(num_experts)]
        self.experts = nn.ModuleList([EfficientExpertGroup(in_channels, out_channels, kernel_size=k) for k in ks])
        self.norm = nn.Sequential(nn.BatchNorm2d(out_channels), nn.SiLU(inplace=True))
        self.register_buffer("load_balancing_loss", torch.tensor(0.0), persistent=False)
        self.register_buffer("expert_usage_counts", torch.zeros(num_experts), persistent=False)
        self.last_routing_snapshot = {}
        self.last_routing_diagnostics = {}
        self.balance_loss_coeff = 1.0
        self._flags = None
        self.last_route = None

    # -- routed-module protocol (ultralytics/nn/modules/routing_protocol.py:34-55) -----------
    @property
    def aux_loss(self):
        return self.load_balancing_loss * float(self.balance_loss_coeff)

    def publish_aux_loss(self, *, step: int, training: bool):
        return self.aux_loss

    def routing_snapshot(self) -> dict:
        return dict(self.last_routing_snapshot)

    def _eager_sparse_enabled(self) -> bool:
        return bool(self.use_sparse_inference and self.use_top_k and self.top_k < self.num_experts)

    def export_capabilities(self) -> dict:
        sparse = self._eager_sparse_enabled()
        return dict(routing_kind="moe", num_experts=self.num_experts, top_k=self.top_k, sparse_dispatch=sparse,
                    eager_sparse_dispatch=sparse, training_sparse_dispatch=False)

    def get_load_balancing_loss(self):
        return self.load_balancing_loss

    def get_expert_usage_stats(self):
        if self.expert_usage_counts.numel() > 0:
            stats = {
                "expert_usage": self.expert_usage_counts.cpu().tolist(),
                "usage_variance": self.expert_usage_counts.var().item(),
                "max_usage": self.expert_usage_counts.max().item(),
                "min_usage": self.expert_usage_counts.min().item(),
            }
            if self.use_top_k:
                stats["active_experts"] = f"{self.top_k}/{self.num_experts}"
                stats["theoretical_speedup"] = f"{self.num_experts / self.top_k:.2f}x"
            return stats
        return None

    def set_top_k(self, top_k):
        if top_k is not None:
            self.top_k = min(top_k, self.num_experts)
            self.routing.top_k = self.top_k
            self.use_top_k = True
            self.routing.use_top_k = True
        else:
            self.top_k = self.num_experts
            self.use_top_k = False
            self.routing.use_top_k = False

    def enable_sparse_inference(self, enable=True):
        self.use_sparse_inference = enable

    # -- packing -------------------------------------------------------------------------
    def _pack(self, dtype, device):
        E, C, Co = self.num_experts, self.in_channels, self.out_channels
        rn = self.routing.routing_network
        hidden = rn[0].out_channels
        ks, dw_parts, dw_off, off = [], [], [], 0
        pw_w = torch.zeros((E, Co, ops.kpad(C)), dtype=torch.float32, device=device)
        pw_b = torch.zeros((E, Co), dtype=torch.float32, device=device)
        for e, ex in enumerate(self.experts):
            cv = ex.conv
            k = cv.depthwise.kernel_size[0]
            if cv.depthwise.stride != (1, 1) or k % 2 == 0 or k > 15:
                raise NotImplementedError("ymk ES_MOE: experts are odd k<=15 stride-1 depthwise + pointwise")
            ks.append(k)
            w = ops.pack_dw_weight(cv.depthwise.weight.detach().float().to(device), dtype)
            dw_parts.append(w.reshape(-1))
            dw_off.append(off)
            off += w.numel()
            wf, bf = ops.fold_bn(cv.pointwise.weight.detach().float().to(device), cv.bn.weight.float().to(device),
                                 cv.bn.bias.float().to(device), cv.bn.running_mean.float().to(device),
                                 cv.bn.running_var.float().to(device), cv.bn.eps)
            pw_w[e, :, :C] = wf.reshape(Co, C)
            pw_b[e] = bf
        ns, nt = _bn_scale_shift(self.norm[0])
        i32 = dict(dtype=torch.int32, device=device)
        return {
            "w1": rn[0].weight.detach().float().reshape(hidden, C).to(device).contiguous(),
            "b1": rn[0].bias.detach().float().to(device).contiguous(),
            "w2": rn[2].weight.detach().float().reshape(E, hidden).to(device).contiguous(),
            "b2": rn[2].bias.detach().float().to(device).contiguous(),
            "dw_w": torch.cat(dw_parts).contiguous(), "dw_off": torch.tensor(dw_off, **i32),
            "ks": torch.tensor(ks, **i32), "kmax": max(ks), "pw_w": pw_w.to(dtype).contiguous(), "pw_b": pw_b.contiguous(),
            "ns": ns.detach().float().to(device).contiguous(), "nt": nt.detach().float().to(device).contiguous(),
        }

    # -- execution -----------------------------------------------------------------------
    def bind_flags(self, flags: torch.Tensor):
        """Share one device flag word (per batch) between all routed layers of a model."""
        self._flags = flags

    def _run(self, x, out=None):
        if x.dim() != 4:
            raise MoERouterError(f"Router input must be 4-D (NCHW), got {x.dim()}-D shape {tuple(x.shape)} "
                                 "[DynamicRoutingLayer]")
        B, H, W, C = x.shape
        if C != self.in_channels:
            raise ShapeMismatchError(expected=f"(N, {self.in_channels}, H, W)", actual=(B, C, H, W),
                                     context="DynamicRoutingLayer")
        pk = self._packed(x.device)
        if self._flags is None or self._flags.device != x.device:
            self._flags = torch.zeros((1,), dtype=torch.int32, device=x.device)
        # Router width and dispatch are separate things (routers.py:477-487 vs modules.py:558-568): the router applies its
        # hard top-k whenever `use_top_k` is set, also when the sparse dispatch is off; the dense forward then sums every
        # expert with those (masked, renormalised) weights -> only the top-k set contributes.  top_k=None: plain softmax.
        top_k = self.top_k if self.use_top_k else self.num_experts
        thr = float(self.dynamic_threshold) if self._eager_sparse_enabled() else -1.0
        route_w, gate_w, sel, csr_off, csr_pair, state = ops.esmoe_route(
            x, pk["w1"], pk["b1"], pk["w2"], pk["b2"], top_k, thr, self._flags)
        cb = self._chunk_images(B, H, W, C, top_k, x.element_size())
        if cb >= B:
            dw = ops.esmoe_dw(x, pk["dw_w"], pk["dw_off"], pk["ks"], pk["kmax"], top_k, sel, csr_off, csr_pair)
            y = ops.esmoe_pw(dw, B, H, W, pk["pw_w"], pk["pw_b"], pk["ns"], pk["nt"], top_k, sel, gate_w, out=out)
        else:
            # The depthwise planes of `cb` images at a time: written by one stage and read by the next while they are still in the
            # 256 MB memory-side cache (all 2 x 64 planes of layer 3 are 840 MB: every byte went out to HBM and came back).  The same
            # kernels on image sub-ranges — per image nothing changes (images are independent through both stages).
            y = out if out is not None else ops.new_act(B, H, W, self.out_channels, x.dtype, x.device)
            for b0 in range(0, B, cb):
                b1 = min(B, b0 + cb)
                dw = ops.esmoe_dw(x[b0:b1], pk["dw_w"], pk["dw_off"], pk["ks"], pk["kmax"], top_k, sel[b0:b1], csr_off, csr_pair)
                ops.esmoe_pw(dw, b1 - b0, H, W, pk["pw_w"], pk["pw_b"], pk["ns"], pk["nt"], top_k, sel[b0:b1], gate_w[b0:b1], out=y[b0:b1])
        # eval-time state the reference keeps (modules.py:706-741), computed by the router's last kernel: views, no arithmetic
        self.expert_usage_counts = state[: self.num_experts]
        self.load_balancing_loss = state[self.num_experts]
        self.last_route = {"route_w": route_w, "gate_w": gate_w, "sel": sel, "csr_off": csr_off, "csr_pair": csr_pair}
        return y

    # depthwise planes kept in flight between the two expert stages (MB); 0 = the whole batch in one pass (yolo_master_amd/options.py)
    chunk_mb = _Opt("moe_chunk_mb")

    def _chunk_images(self, B, H, W, C, top_k, es):
        if self.chunk_mb <= 0:
            return B
        per_image = top_k * H * W * C * es
        return max(1, min(B, int(self.chunk_mb * 1e6 // per_image)))

    def check_flags(self):
        """Host-side check of the device flag word (one sync); raises the reference's exception types."""
        if self._flags is None:
            return
        f = int(self._flags.item())
        self.last_routing_diagnostics = {
            "all_finite": (f & 3) == 0,
            "first_nonfinite_boundary": "router_input" if f & 1 else "router_logits" if f & 2 else None,
        }
        if f & 1:
            self._flags.zero_()
            raise MoERouterError("Router input contains NaN/Inf values [DynamicRoutingLayer]")
        if f & 2:
            self._flags.zero_()
            raise MoERouterError("DynamicRoutingLayer internal output contains NaN/Inf values")

    def forward(self, x):
        if self.training:
            raise RuntimeError("ES_MOE: the ymk path implements eval-mode inference only")
        if x.dim() != 4:
            raise MoERouterError(f"Router input must be 4-D (NCHW), got {x.dim()}-D shape {tuple(x.shape)} "
                                 "[DynamicRoutingLayer]")
        if x.shape[1] != self.in_channels:
            raise ShapeMismatchError(expected=f"(N, {self.in_channels}, H, W)", actual=tuple(x.shape),
                                     context="DynamicRoutingLayer")
        y = self._run(to_nhwc(x, self.ymk_dtype))
        self.check_flags()  # module-level API keeps the reference's eager raise-on-NaN contract
        return y.permute(0, 3, 1, 2)
