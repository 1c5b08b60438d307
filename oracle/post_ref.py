"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the step right after the hot path: `scale_boxes` / `clip_boxes`, which map
the detections from the letterboxed network input back to the original image (`ultralytics/utils/ops.py:119-174, 176-205`,
called per image by `DetectionPredictor.construct_result`, `models/yolo/detect/predict.py:109-122`), and `process_mask` /
`crop_mask` of the segmentation predictor (`utils/ops.py:477-528`).
Pinned against the real reference by `tests/golden/make_golden_post.py` (bit-exact), checked without it by
`tests/test_oracle_post.py`."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

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


def crop_mask(masks: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
    """utils/ops.py:477-497: zero everything outside [x1, x2) x [y1, y2) (boxes in the masks' pixel coordinates)."""
    _, h, w = masks.shape
    x1, y1, x2, y2 = torch.chunk(boxes[:, :, None], 4, 1)
    r = torch.arange(w, dtype=x1.dtype)[None, None, :]
    c = torch.arange(h, dtype=x1.dtype)[None, :, None]
    return masks * ((r >= x1) * (r < x2)) * ((c >= y1) * (c < y2))


def process_mask(protos: torch.Tensor, masks_in: torch.Tensor, bboxes: torch.Tensor, shape, upsample: bool = False) -> torch.Tensor:
    """utils/ops.py:500-528: coefficient x prototype product, then either bilinear upsampling to `shape` and a crop with the
    boxes (upsample=True) or a crop at prototype resolution with the boxes scaled down; binarised at 0.  protos [nm, mh, mw]."""
    c, mh, mw = protos.shape
    if masks_in.shape[0] == 0:
        return torch.zeros((0, *(shape if upsample else (mh, mw))), dtype=torch.uint8)
    masks = (masks_in @ protos.float().view(c, -1)).view(-1, mh, mw)
    if upsample:
        masks = crop_mask(F.interpolate(masks[None], shape, mode="bilinear")[0], bboxes)
    else:
        ratios = torch.tensor([[mw / shape[1], mh / shape[0], mw / shape[1], mh / shape[0]]])
        masks = crop_mask(masks, bboxes * ratios)
    return masks.gt(0.0).byte()


# ---------------------------------------------------------------------------------- validation matching
def box_iou(box1: np.ndarray, box2: np.ndarray, eps: float = 1e-7) -> np.ndarray:
    """utils/metrics.py:82-104 in numpy fp32, same operation order: inter / (area1 + area2 - inter + eps)."""
    a, b = np.asarray(box1, f32)[:, None, :4], np.asarray(box2, f32)[None, :, :4]
    wh = np.clip(np.minimum(a[..., 2:], b[..., 2:]) - np.maximum(a[..., :2], b[..., :2]), 0, None).astype(f32)
    inter = wh[..., 0] * wh[..., 1]
    area1 = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area2 = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    return (inter / (area1 + area2 - inter + f32(eps))).astype(f32)


def match_predictions(pred_classes: np.ndarray, true_classes: np.ndarray, iou: np.ndarray, iouv: np.ndarray) -> np.ndarray:
    """BaseValidator.match_predictions (engine/validator.py:301-336), numpy path, restated line by line.  iou: [L labels, D detections].
    Returns correct bool [D, T]."""
    correct = np.zeros((pred_classes.shape[0], iouv.shape[0]), bool)
    correct_class = true_classes[:, None] == pred_classes
    iou = (np.asarray(iou, f32) * correct_class).astype(f32)
    for i, threshold in enumerate(np.asarray(iouv, f32).tolist()):
        matches = np.array(np.nonzero(iou >= f32(threshold))).T
        if matches.shape[0]:
            if matches.shape[0] > 1:
                matches = matches[iou[matches[:, 0], matches[:, 1]].argsort()[::-1]]
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
            correct[matches[:, 1].astype(int), i] = True
    return correct


def batch_stats(dets_per_image, labels_per_image, iouv):
    """DetectionValidator.update_metrics' per-image record (models/yolo/detect/val.py:176-192 + _process_batch :313-327), restated:
    dets [n, 6] (xyxy, conf, cls), labels [L, 5] (cls, xyxy).  Returns the list of dicts handed to `metrics.update_stats`."""
    out = []
    for d, l in zip(dets_per_image, labels_per_image):
        d, l = np.asarray(d, f32).reshape(-1, 6), np.asarray(l, f32).reshape(-1, 5)
        if l.shape[0] == 0 or d.shape[0] == 0:
            tp = np.zeros((d.shape[0], len(iouv)), bool)
        else:
            tp = match_predictions(d[:, 5], l[:, 0], box_iou(l[:, 1:], d[:, :4]), iouv)
        no_pred = d.shape[0] == 0
        out.append({"tp": tp, "target_cls": l[:, 0], "target_img": np.unique(l[:, 0]), "conf": np.zeros(0) if no_pred else d[:, 4],
                    "pred_cls": np.zeros(0) if no_pred else d[:, 5]})
    return out
