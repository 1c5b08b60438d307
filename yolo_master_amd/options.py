"""Host-side switches of the product, in ONE place.

Every switch is a named field of `OPTIONS` with the default the measurements chose; the environment (`YMK_DISABLE` / `YMK_ENABLE` bit masks,
`YMK_MOE_CHUNK_MB`) is read ONCE, here, when the package is imported — the masks exist so that a profiling script can time the same process
image with and without one specialised kernel (tools/micro/env_ab.sh, `tools/micro/calls_ab.sh`); every combination computes the same result
(tests/test_gpu_variants.py).  Tests and callers set fields directly (`OPTIONS.fused_mlp = False`), they never touch the environment.

The C side has its own, smaller set (csrc/ymk_common.h `ymk_disabled()` / `ymk_enabled()`: kernel selection INSIDE an entry point); the bits
below marked (C) are read there as well, from the same variables.

    field                   default   YMK_DISABLE bit      what it selects
    res_prefetch            on        16 (C)               register prefetch of residual operands in the spatial-tile 3x3 kernel (kernel-name tag only here)
    expert_conv_glds        on        512                  routed experts of the gated family on the LDS-DMA core (-> per-expert convolutions)
    fused_mlp               on        1024                 ABlock `x + mlp(x)` as one kernel (csrc/mlp.hip) (-> two 1x1 convolutions)
    fused_stem_pair         on        2048                 YAML rows 0 + 1 as one kernel (csrc/stem2.hip) (-> stem kernel + 3x3 convolution)
    fused_c3k2              on        8192                 YAML row 2 as one kernel (csrc/c3k2f.hip) (-> its four convolutions)
    fused_detect_cls        on        16384                Detect class branch of a level as one kernel (csrc/detcls.hip) (-> five convolutions)
    fused_decode            on        4194304              DFL decode / sigmoid in the producers' epilogues (-> fp32 logits + detect_decode kernel)
    fused_qkv_attn          on        2097152 (C)          AAttn's qkv 1x1 + attention as ONE kernel where C = heads * 32 <= 128 (csrc/attn.hip area_attn_qkv_kernel): K / V^T are
                                                           produced in LDS, only v and the attention output are stored (-> 1x1 convolution + ymk_area_attn)
    pooled_producers        on        1048576              the streaming 1x1 that produces an ES-MoE layer's input also leaves its per-tile channel sums
                                                           (ymk_conv1x1_pooled): the router pools those instead of re-reading the map (-> plain convolution)

    field                   default   YMK_ENABLE bit       what it selects
    detect_level_streams    off       16                   a side HIP stream per Detect level (measured slower: profiles/r03_negative_results.txt)
    detect_early_levels     off       32                   Detect levels launched as soon as their input map exists (no gain on top of the batch pipeline)
    detect_keep_raw         off       256                  fused decode that ALSO materialises the fp32 logits (`preds["raw"]` without recomputation)

    moe_chunk_mb            0         YMK_MOE_CHUNK_MB     ES-MoE expert stages walked in image chunks whose depthwise planes total at most this many MB
                                                           (0: the whole batch per stage)

C-side, read by libymk itself at first use (A/B switches of one kernel's launch shape; every value computes the same result):
    YMK_NMS_GREEDY_WAVES    16        waves per image in the greedy NMS pass (4 = the rounds 1-4 form)            csrc/nms.hip
    YMK_ATTN_WAVES          0         5 | 6 forces five- / six-wave workgroups in the resident area attention      csrc/attn.hip
    YMK_WS_MIN_TILES, YMK_GLDS_MIN_TILES, YMK_GLDS_MIN_K, YMK_GLDS_BIG_MIN_TILES, YMK_GLDS_SMALL_BELOW, YMK_GLDS_TILE, YMK_GLDS_TAP_OUTER,
    YMK_GLDS_THREE_STAGE    —         kernel-selection thresholds of the convolution cores                         csrc/conv.hip, conv_glds.hip
    YMK_DW_WG_TARGET        0         tiles per workgroup of the depthwise stencil                                 csrc/dwconv.hip
"""
from __future__ import annotations

import os
from dataclasses import dataclass


def _mask(name: str) -> int:
    try:
        return int(os.environ.get(name, "0"), 0)
    except ValueError:
        return 0


@dataclass
class Options:
    res_prefetch: bool = True
    expert_conv_glds: bool = True
    fused_mlp: bool = True
    fused_stem_pair: bool = True
    fused_c3k2: bool = True
    fused_detect_cls: bool = True
    fused_decode: bool = True
    pooled_producers: bool = True
    fused_qkv_attn: bool = True
    detect_level_streams: bool = False
    detect_early_levels: bool = False
    detect_keep_raw: bool = False
    moe_chunk_mb: float = 0.0
    disable_mask: int = 0      # the raw masks, for code that reports them (bench.py)
    enable_mask: int = 0

    @classmethod
    def from_env(cls) -> "Options":
        d, e = _mask("YMK_DISABLE"), _mask("YMK_ENABLE")
        try:
            chunk = float(os.environ.get("YMK_MOE_CHUNK_MB", "0"))
        except ValueError:
            chunk = 0.0
        return cls(res_prefetch=not d & 16, expert_conv_glds=not d & 512, fused_mlp=not d & 1024, fused_stem_pair=not d & 2048,
                   fused_c3k2=not d & 8192, fused_detect_cls=not d & 16384, fused_decode=not d & 4194304, pooled_producers=not d & 1048576, fused_qkv_attn=not d & 2097152,
                   detect_level_streams=bool(e & 16), detect_early_levels=bool(e & 32), detect_keep_raw=bool(e & 256),
                   moe_chunk_mb=chunk, disable_mask=d, enable_mask=e)


OPTIONS = Options.from_env()
