"""Device-side pre-processing with the reference predictor's semantics: `LetterBox` (ultralytics/data/augment.py:1646-1830) +
`BasePredictor.preprocess` (engine/predictor.py:155-178) for a batch of uint8 HWC BGR images, one libymk launch
(`ymk_letterbox_preprocess`, include/ymk_next.h).  The geometry is `LetterBox.get_params` evaluated on the host (Python
arithmetic, Python `round`); resize, padding, channel swap, layout change and the /255 run on the GPU.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def letterbox_params(shape, new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True, stride=32) -> dict:
    """LetterBox.get_params (augment.py:1752-1800) for an image of `shape` = (h, w): new_unpad (w, h), top / bottom / left / right."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = round(shape[1] * r), round(shape[0] * r)
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    elif scale_fill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    if center:
        dw /= 2
        dh /= 2
    top, bottom = (round(dh - 0.1) if center else 0), round(dh + 0.1)
    left, right = (round(dw - 0.1) if center else 0), round(dw + 0.1)
    return {"new_unpad": (int(new_unpad[0]), int(new_unpad[1])), "top": int(top), "bottom": int(bottom), "left": int(left),
            "right": int(right), "ratio": (float(ratio[0]), float(ratio[1]))}


def preprocess(images, imgsz=(640, 640), device="cuda:0", auto=False, scale_fill=False, scaleup=True, center=True, stride=32,
               pad_value=114, bgr=True) -> torch.Tensor:
    """images: list of uint8 HWC 3-channel arrays / tensors (BGR as cv2.imread gives them; host or device).  Returns the network
    input fp32 [B, 3, H, W] on `device`, RGB, 0..1.  All images must letterbox to the same (H, W) (as `np.stack` in the reference
    requires): always true with auto=False."""
    if isinstance(imgsz, int):
        imgsz = (imgsz, imgsz)
    dev = torch.device(device)
    ops.require_gpu(torch.empty(0, device=dev), "yolo_master_amd pre-processing")
    geoms, offs, parts, off, out_hw = [], [], [], 0, None
    for im in images:
        t = im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im))
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("preprocess: uint8 HWC images with 3 channels")
        h, w = int(t.shape[0]), int(t.shape[1])
        p = letterbox_params((h, w), imgsz, auto, scale_fill, scaleup, center, stride)
        nw, nh = p["new_unpad"]
        hw = (nh + p["top"] + p["bottom"], nw + p["left"] + p["right"])
        if out_hw is None:
            out_hw = hw
        elif hw != out_hw:
            raise ValueError(f"preprocess: images letterbox to different shapes {out_hw} vs {hw} (use auto=False)")
        geoms.append([h, w, nh, nw, p["top"], p["left"]])
        offs.append(off)
        parts.append(t.contiguous().reshape(-1))
        off += h * w * 3
    B = len(parts)
    if B == 0:
        raise ValueError("preprocess: empty batch")
    src = torch.cat([p_.to(dev, non_blocking=True) for p_ in parts])
    geom = torch.tensor(geoms, dtype=torch.int32).to(dev, non_blocking=True)
    offd = torch.tensor(offs, dtype=torch.int64).to(dev, non_blocking=True)
    H, W = out_hw
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    e0 = ops.TIMER.begin()
    check(lib.ymk_letterbox_preprocess(ops._p(src), ops._p(offd), ops._p(geom), ops._p(out), B, H, W, int(pad_value), int(bool(bgr)),
                                       ops._stream()), "letterbox_preprocess")
    ops.TIMER.end(e0, "preprocess", src.numel() + out.numel() * 4, 0, f"{B} images -> {H}x{W}")
    return out
