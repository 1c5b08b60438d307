"""ORACLE (test infrastructure — never imported by the product path).

numpy restatement of the predictor's pre-processing: LetterBox (ultralytics/data/augment.py:1646-1830) followed by
BasePredictor.preprocess (engine/predictor.py:155-178).

* `letterbox_params` restates LetterBox.get_params (augment.py:1752-1800) — pure Python arithmetic, Python round() included.
  PINNED: tests/golden/pre_params.json holds the real reference's get_params output for a sweep of shapes
  (tests/golden/make_golden_pre.py imports the reference class; get_params never touches cv2).
* `resize_linear_u8` restates what `cv2.resize(img, new_unpad, interpolation=cv2.INTER_LINEAR)` (augment.py:1807) computes on
  uint8 images.  cv2 is a third-party dependency of the reference that is NOT vendored in /root/reference and not installed in
  this environment (pyproject.toml: "opencv-python>=4.6.0"); the algorithm restated here is OpenCV 4.x's generic C++ path,
  modules/imgproc/src/resize.cpp: 11-bit fixed-point coefficients (INTER_RESIZE_COEF_BITS = 11), source position
  (d + 0.5) * scale - 0.5 evaluated in double and narrowed to float, clamped taps, HResizeLinear to int, VResizeLinear
  `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`, and the exact-2x-downscale case that `resize()` re-routes
  to the 2x2 area average.  PARITY UNPINNED for this function (no cv2 to run; IPP-enabled OpenCV builds may differ by one
  level); what pins it here are its invariants (tests/test_oracle_pre.py): identity, constants, monotone ramps, and agreement
  within one grey level with torch's float bilinear (align_corners=False), which is the same sampling geometry.
* `preprocess` = stack, BGR->RGB, HWC->CHW, float32, /255 exactly as predictor.py:166-177 (numpy / torch semantics).
"""
from __future__ import annotations

import numpy as np


def letterbox_params(shape, new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True, stride=32):
    """LetterBox.get_params: shape = (h, w) of the image -> dict(new_unpad=(w, h), top, bottom, left, right, ratio)."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = round(shape[1] * r), round(shape[0] * r)
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
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


def _coef(dsize: int, ssize: int, vertical: bool = False):
    """Per destination index: the two taps and the two 11-bit coefficients (OpenCV resize.cpp, linear kernel).  Horizontal: at the
    borders the tap moves inside and the fraction is zeroed (`if (sx < 0) fx = 0, sx = 0; if (sx >= ssize.width - 1) fx = 0, sx =
    ssize.width - 1`).  Vertical: the fraction is kept and resizeGeneric_Invoker clamps the ROW INDICES (`clip(yofs[dy] + k, 0,
    ssize.height)`), so with both rows equal the two products round separately (first / last rows of upscaled images can be one LSB
    lower than with a zeroed fraction)."""
    scale = 1.0 / (dsize / ssize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = f - s.astype(np.float32)
    if vertical:
        s0, s1 = np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1)
    else:
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= ssize - 1
        f[hi], s[hi] = 0.0, ssize - 1
        s0, s1 = s, np.minimum(s + 1, ssize - 1)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int32)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int32)
    return s0, s1, a0, a1


def resize_linear_u8(img: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(img, dsize=(w, h), interpolation=cv2.INTER_LINEAR) for uint8 HWC images (see module docstring)."""
    img = np.asarray(img, np.uint8)
    sh, sw = img.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dh, dw) == (sh, sw):
        return img.copy()
    if sh == 2 * dh and sw == 2 * dw:   # exact 2x downscale -> 2x2 area average, rounded
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, sx1, ax0, ax1 = _coef(dw, sw)
    sy, sy1, by0, by1 = _coef(dh, sh, vertical=True)
    s = img.astype(np.int32)
    h = s[:, sx] * ax0[None, :, None] + s[:, sx1] * ax1[None, :, None]          # [sh, dw, c], scale 2^11
    h0, h1 = h[sy], h[sy1]
    out = (((by0[:, None, None] * (h0 >> 4)) >> 16) + ((by1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img: np.ndarray, new_shape=(640, 640), pad_value=114, **kw) -> np.ndarray:
    """LetterBox.__call__(image=img) for a 3-channel uint8 image (apply_image, augment.py:1802-1826)."""
    p = letterbox_params(img.shape[:2], new_shape, **kw)
    if tuple(img.shape[:2][::-1]) != p["new_unpad"]:
        img = resize_linear_u8(img, p["new_unpad"])
    h, w = img.shape[:2]
    out = np.full((h + p["top"] + p["bottom"], w + p["left"] + p["right"], 3), pad_value, np.uint8)
    out[p["top"]: p["top"] + h, p["left"]: p["left"] + w] = img
    return out


def preprocess(images, new_shape=(640, 640), **kw) -> np.ndarray:
    """BasePredictor.preprocess for a list of BGR uint8 images: letterbox each, stack, BGR->RGB, BHWC->BCHW, float32, /255."""
    im = np.stack([letterbox(x, new_shape, **kw) for x in images])
    im = np.ascontiguousarray(im[..., ::-1].transpose((0, 3, 1, 2)))
    return im.astype(np.float32) / np.float32(255)
