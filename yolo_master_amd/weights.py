"""Deterministic synthetic weights for parity tests and benchmarks.

No checkpoints or datasets are reachable offline, so both sides of every
parity test load the same seeded ``state_dict``.  The recipe is keyed by parameter *name*
(crc32) so it is independent of construction order, and it deliberately exercises what the
reference's default init leaves dormant: non-identity BatchNorm statistics
(so the BN fold matters), router logits that differ per image (so the sparse dispatch and
the 0.4 threshold see both outcomes), and a Detect class bias high enough that NMS receives
~10^3 candidates per image.
"""
from __future__ import annotations

import re
import zlib
from pathlib import Path

import numpy as np
import torch

CFG_DIR = Path(__file__).resolve().parent / "cfg"


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))
    return g


_DETECT_TAIL_BIAS = re.compile(r"\.(one2one_)?cv[23]\.\d+\.2\.bias$")


def synth_state_dict(template: dict, seed: int = 0, router_scale: float = 4.0, cls_bias_shift: float = 6.0,
                     conv_gain: float = 1.0, tail_gain: float = 1.2, calib: str | None = "auto") -> dict:
    """Return a new state_dict with the template's keys/shapes/dtypes and seeded values.

    ``calib``: path of a ``bn_calib_*.npz`` (tools/make_calibration.py) whose BatchNorm running statistics
    replace the random ones; "auto" picks yolo_master_amd/cfg/bn_calib_<scale>.npz by matching shapes
    (seed 0 only — the statistics belong to the seed-0 weights); None keeps the random statistics."""
    out = {}
    for name, t in template.items():
        g = _gen(seed, name)
        shp = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            v = torch.zeros(shp, dtype=t.dtype)
        elif name.endswith("dfl.conv.weight"):
            v = torch.arange(shp[1], dtype=torch.float32).view(shp)  # fixed DFL integral weights
        elif name.endswith("running_mean"):
            v = torch.randn(shp, generator=g) * 0.1
        elif name.endswith("running_var"):
            v = torch.rand(shp, generator=g) + 0.5
        elif ".bn." in name or ".norm.0." in name:
            v = (torch.rand(shp, generator=g) * 0.4 + 0.8) if name.endswith("weight") else torch.randn(shp, generator=g) * 0.1
        elif "routing_network" in name:
            if name.endswith("weight"):
                fan_in = shp[1]
                v = torch.randn(shp, generator=g) / fan_in**0.5
                if ".routing_network.2." in name:
                    v = v * router_scale
            else:
                v = torch.randn(shp, generator=g) * (0.5 if ".routing_network.2." in name else 0.1)
        elif name.endswith("weight") and len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            if ".attn." in name or ".mlp." in name:  # ABlock convs: trunc_normal(std=0.02) scale (block.py:1776-1785)
                v = torch.randn(shp, generator=g).clamp_(-2, 2) * 0.02 * (4.0 if ".pe." in name else 1.0)
            else:
                v = torch.randn(shp, generator=g) * (conv_gain * (1.0 / fan_in) ** 0.5)
                if name.endswith((".cv2.0.2.weight", ".cv2.1.2.weight", ".cv2.2.2.weight", ".cv3.0.2.weight",
                                  ".cv3.1.2.weight", ".cv3.2.2.weight")):
                    v = v * tail_gain  # Detect tail 1x1: spread the box/cls logits (std ~1-2)
        elif name.endswith("bias") and _DETECT_TAIL_BIAS.search(name):
            # Detect tail convs: keep the reference's bias_init value (the template's: deterministic, head.py bias_init) + jitter
            v = t.detach().clone().float() + torch.randn(shp, generator=g) * 0.1
            if ".cv3." in name:
                v = v + cls_bias_shift
        elif name.endswith("bias"):
            # any other plain convolution bias (Segment's mask-coefficient tail, Proto's transposed convolution ...): the template holds
            # torch's RANDOM initialisation there — two models built in one process would get different values — so it is never read
            v = torch.randn(shp, generator=g) * 0.1
        elif name.endswith("gamma"):
            v = torch.full(shp, 0.01) + torch.randn(shp, generator=g) * 0.001
        else:
            v = torch.randn(shp, generator=g) * 0.02
        out[name] = v.to(t.dtype)
    if calib == "auto":
        calib = None
        if seed == 0:
            for f in sorted(CFG_DIR.glob("bn_calib_*.npz")):
                z = np.load(f)
                if all(k in out and tuple(out[k].shape) == z[k].shape for k in z.files) and \
                        sum(k.endswith("running_mean") for k in out) == len(z.files) // 2:
                    calib = f
                    break
    if calib is not None:
        z = np.load(calib)
        for k in z.files:
            out[k] = torch.from_numpy(z[k]).to(out[k].dtype)
    return out


def expert_imbalance(sd: dict, alpha_image: float, alpha_token: float) -> dict:
    """The expert-imbalance knob of BASELINE config 5: a copy of `sd` with expert 0's logit raised in every router —
    the ES-MoE routers' second layer (`routing_network.2.bias`, moe/routers.py:429-457), the gated blocks' local stream
    (`routing.local_conv.6.bias`, moe/gated.py:124-166; the blend scales it by 1 - sigmoid(alpha_param)) by `alpha_image`, and the
    per-token MoT / MoA routers (`router.router.3.bias`, mot/router.py:243-295, moa/router.py:29-62) by `alpha_token`."""
    out = {k: v.clone() for k, v in sd.items()}
    for k in out:
        if k.endswith("routing.local_conv.6.bias") or k.endswith("routing_network.2.bias"):
            out[k][0] += alpha_image
        elif k.endswith("router.router.3.bias"):
            out[k][0] += alpha_token
    return out


def synth_input(B: int, H: int = 640, W: int = 640, seed: int = 1) -> torch.Tensor:
    """Seeded synthetic images in [0,1], NCHW fp32 (a tensor source skips letterbox and /255,
    engine/predictor.py:164-177).  Every image gets its own tint / contrast / blocky low-frequency
    layout so that pooled router features (and hence expert choices) differ between images."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    noise = torch.rand((B, 3, H, W), generator=g)
    gain = torch.rand((B, 1, 1, 1), generator=g) * 0.4 + 0.6
    tint = torch.rand((B, 3, 1, 1), generator=g) * 0.2
    gh, gw = max(H // 80, 1), max(W // 80, 1)
    blocks = torch.rand((B, 3, gh, gw), generator=g)
    blocks = torch.nn.functional.interpolate(blocks, size=(H, W), mode="nearest")
    return (tint + gain * (0.5 * noise + 0.5 * blocks)).clamp_(0.0, 1.0)
