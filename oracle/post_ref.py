"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the step right after the hot path: `scale_boxes` / `clip_boxes`, which map
the detections from the letterboxed network input back to the original image (`ultralytics/utils/ops.py:119-174, 176-205`,
called per image by `DetectionPredictor.construct_result`, `models/yolo/detect/predict.py:109-122`).
Pinned against the real reference by `tests/golden/make_golden_post.py` (bit-exact), checked without it by
`tests/test_oracle_post.py`."""
from __future__ import annotations

import numpy as np

f32 = np.float32


def letterbox_params(img1_shape, img0_shape, ratio_pad=None):
    """(gain, pad_x, pad_y) exactly as scale_boxes derives them (utils/ops.py:141-147): Python double arithmetic and
    Python's round() (half to even)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad_x = round((img1_shape[1] - round(img0_shape[1] * gain)) / 2 - 0.1)
        pad_y = round((img1_shape[0] - round(img0_shape[0] * gain)) / 2 - 0.1)
    else:
        gain = ratio_pad[0][0]
        pad_x, pad_y = ratio_pad[1]
    return gain, pad_x, pad_y


def scale_boxes(img1_shape, boxes: np.ndarray, img0_shape, ratio_pad=None, padding=True, xywh=False) -> np.ndarray:
    """utils/ops.py:119-174 on an fp32 [N, >=4] array (returns a copy): subtract the padding, divide by the gain (the
    gain rounded to fp32 first, as torch does for a Python scalar operand), clip xyxy boxes to the original image."""
    gain, pad_x, pad_y = letterbox_params(img1_shape, img0_shape, ratio_pad)
    b = np.array(boxes, dtype=f32, copy=True)
    if padding:
        b[..., 0] -= f32(pad_x)
        b[..., 1] -= f32(pad_y)
        if not xywh:
            b[..., 2] -= f32(pad_x)
            b[..., 3] -= f32(pad_y)
    b[..., :4] /= f32(gain)
    if xywh:
        return b
    h, w = img0_shape[:2]
    b[..., 0] = np.clip(b[..., 0], 0, w)
    b[..., 1] = np.clip(b[..., 1], 0, h)
    b[..., 2] = np.clip(b[..., 2], 0, w)
    b[..., 3] = np.clip(b[..., 3], 0, h)
    return b
