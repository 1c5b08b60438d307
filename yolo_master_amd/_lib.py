"""ctypes binding of libymk.so — the C-ABI boundary (include/ymk.h).

The library is hand-written HIP for gfx950; there is no CPU or PyTorch fallback.  If the
shared object is missing or a symbol declared in include/ymk.h is not exported, importing
an op raises immediately (``YmkLibraryError``) instead of silently degrading.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

__all__ = ["lib", "load", "YmkLibraryError", "ConvDesc", "check", "LIB_PATH", "SYMBOLS", "SYMBOLS_MIXTURE", "SYMBOLS_NEXT"]

LIB_PATH = Path(__file__).resolve().parent / "libymk.so"
LIB_F16_PATH = Path(__file__).resolve().parent / "libymk_f16.so"   # the same sources / ABI, 16-bit element format = IEEE binary16

YMK_F32, YMK_H16 = 0, 1
YMK_BF16 = YMK_H16            # include/ymk.h: the 16-bit element type of the loaded build (bf16 in libymk.so, fp16 in libymk_f16.so)
H16_FORMAT_BF16, H16_FORMAT_F16 = 1, 2
ACT_NONE, ACT_SILU = 0, 1
FLAG_NONFINITE_INPUT, FLAG_NONFINITE_LOGITS, FLAG_NMS_OVERFLOW = 1, 2, 4


class YmkLibraryError(RuntimeError):
    """libymk.so is missing / stale; the MI355X-native path cannot run."""


class ConvDesc(C.Structure):
    """Mirror of ``ymk_conv_desc`` (include/ymk.h)."""

    _fields_ = [
        ("dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32),
        ("ldx", C.c_int32), ("ldy", C.c_int32), ("ldr", C.c_int32),
        ("Kpad", C.c_int32), ("act", C.c_int32),
    ]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every function declared in include/ymk.h
ABI_VERSION = 4   # include/ymk.h YMK_ABI_VERSION

SYMBOLS = {
    "ymk_abi_version": (C.c_int, []),
    "ymk_build_info": (C.c_char_p, []),
    "ymk_h16_format": (C.c_int, []),
    "ymk_conv2d": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "ymk_conv2d_last_variant": (_i32, []),
    "ymk_conv1x1_cat2": (C.c_int, [C.POINTER(ConvDesc), _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "ymk_conv2d_stem_nchw": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_dwconv2d": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_esmoe_route_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "ymk_esmoe_route": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ymk_esmoe_dw": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ymk_esmoe_pw": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp,
                               _vp, _i32, _vp]),
    "ymk_stem_pair_supported": (C.c_int, [_i32] * 8),
    "ymk_stem_pair": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp]),
    "ymk_c3k2_fused_supported": (C.c_int, [_i32] * 7),
    "ymk_c3k2_fused": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "ymk_conv1x1_pool_chunks": (_i32, [_vp]),
    "ymk_conv1x1_pooled": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ymk_pool_tiles128": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ymk_c3k2_fused_pool_chunks": (_i32, [_i32, _i32]),
    "ymk_c3k2_fused_pooled": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "ymk_esmoe_route_pooled": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ymk_detect_cls_fused_supported": (C.c_int, [_i32] * 4),
    "ymk_detect_cls_fused": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp,
                                       _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ymk_detect_box_tail_supported": (C.c_int, [_i32, _i32, _i32, _i32]),
    "ymk_detect_box_tail": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp]),
    "ymk_mlp_fused_supported": (C.c_int, [_i32, _i32, _i32]),
    "ymk_mlp_fused": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "ymk_area_attn": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_area_attn_qkv_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "ymk_area_attn_qkv": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_upsample2x": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_copy_channels": (C.c_int, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "ymk_scale_residual": (C.c_int, [_i32, _vp, _vp, _vp, _vp, C.c_int64, _i32, _i32, _i32, _i32, _vp]),
    "ymk_nhwc_to_nchw_f32": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ymk_detect_decode": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp]),
    "ymk_nms_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "ymk_nms_batched": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _sz, _vp]),
    "ymk_nms_gather_rows": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "ymk_cw_refine": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _sz, _vp]),
}

# Config-5 rows (include/ymk_mixture.h): first implementation, compiled but not yet run on hardware — bound separately so
# that the validated table above stays exactly the surface of include/ymk.h.
SYMBOLS_MIXTURE = {
    "ymk_activation": (C.c_int, [_i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    "ymk_group_norm": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _i32,
                                 _vp, _vp]),
    "ymk_layer_norm": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i64, _i32, _vp, _vp, _f32, _vp]),
    "ymk_eltwise": (C.c_int, [_i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _f32, _vp]),
    "ymk_fma_gate": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ymk_channel_gate": (C.c_int, [_i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ymk_batch_scale": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _f32, _f32, _vp]),
    "ymk_weighted_sum": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ymk_mean_upsampled": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp, _i32,
                                     _i32, _i32, _i32, _i32, _vp]),
    "ymk_adaptive_avg_pool": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_avg_pool": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_channel_stats": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ymk_token_softmax": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "ymk_pooled_softmax_route": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ymk_scene_workspace_bytes": (_sz, [_i32, _i32]),
    "ymk_scene_bias": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp,
                                 _sz, _vp]),
    "ymk_moa_sparse_gate": (C.c_int, [_vp, _i32, _i64, _i32, _f32, _vp, _vp, _i32, _vp, _vp]),
    "ymk_gated_route_decide": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ymk_expert_gather": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ymk_expert_dw3": (C.c_int, [_i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ymk_channel_shuffle_cat": (C.c_int, [_i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _i64, _vp]),
    "ymk_attention": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "ymk_window_attention": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32,
                                       _i32, _vp, _vp, _vp, _vp]),
    "ymk_linear_attention": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ymk_deform_attention": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                       _vp]),
}
# include/ymk_next.h: opt-in entry points outside the validated surface
SYMBOLS_NEXT = {
    "ymk_conv2d_glds": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ymk_conv1x1_cat2_glds": (C.c_int, [C.POINTER(ConvDesc), _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "ymk_pixel_shuffle2": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_tokens_to_rows": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ymk_mask_coeff_gather": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "ymk_process_mask": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp]),
    "ymk_expert_conv_glds": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "ymk_scale_boxes": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ymk_box_iou": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _f32, _vp, _vp]),
    "ymk_match_predictions_workspace_bytes": (_sz, [_i32, _i32]),
    "ymk_match_predictions": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _f32, _vp, _vp, _sz, _vp]),
    "ymk_letterbox_preprocess": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
}
ACT_SIGMOID, ACT_GELU = 2, 3
ELT_MUL, ELT_SIGMOID_MUL, ELT_LERP, ELT_CLAMP_ADD = 0, 1, 2, 3

_lib = None
_lib_f16 = None


def load_f16() -> C.CDLL:
    """libymk_f16.so: the fp16 build of the same ABI (torch.float16 tensors); loaded on first use, RTLD_LOCAL like libymk.so, so the
    two builds' identically named symbols never meet."""
    global _lib_f16
    if _lib_f16 is None:
        h = load(LIB_F16_PATH)
        if h.ymk_h16_format() != H16_FORMAT_F16:
            raise YmkLibraryError(f"{LIB_F16_PATH} is not an fp16 build (ymk_h16_format() = {h.ymk_h16_format()})")
        _lib_f16 = h
    return _lib_f16


def load(path: os.PathLike | None = None) -> C.CDLL:
    """Load libymk.so (once) and bind every C-ABI symbol; raise loudly when unavailable."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise YmkLibraryError(
            f"{p} not found: build it with `python -m yolo_master_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU/PyTorch fallback for the ymk forward path."
        )
    # torch ships its own libamdhip64.so.7; import it first so both share one HIP runtime
    import torch  # noqa: F401

    try:
        h = C.CDLL(str(p))
    except OSError as e:  # pragma: no cover
        raise YmkLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise YmkLibraryError(f"{p} does not export {name}; rebuild libymk") from e
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in {**SYMBOLS_MIXTURE, **SYMBOLS_NEXT}.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise YmkLibraryError(f"{p} does not export {name}; rebuild libymk") from e
        fn.restype, fn.argtypes = res, args
    if h.ymk_abi_version() != ABI_VERSION:
        raise YmkLibraryError(f"{p}: ABI version {h.ymk_abi_version()} != {ABI_VERSION} (stale build: run python -m yolo_master_amd.build)")
    if path is None:
        _lib = h
    return h


class _Lazy:
    def __getattr__(self, k):
        return getattr(load(), k)


lib = _Lazy()

_ERR = {-1: "YMK_E_BADARG (unsupported shape/alignment/dtype)", -2: "YMK_E_LAUNCH", -3: "YMK_E_WORKSPACE"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"libymk {what} failed: {_ERR.get(rc, rc)}")
