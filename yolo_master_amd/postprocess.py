"""The step right after the hot path: detections back to original-image coordinates.

``scale_boxes`` mirrors ``ultralytics.utils.ops.scale_boxes`` (utils/ops.py:119-174: same arguments, in place, returns the
tensor); ``scale_detections`` is the batched form over the padded output of ``nms_padded`` (what
``DetectionPredictor.construct_result`` does per image, models/yolo/detect/predict.py:109-122).  The letterbox parameters
are computed on the host with the reference's own arithmetic (Python doubles, round-half-even); the arithmetic on the boxes
runs in libymk (``ymk_scale_boxes``, include/ymk_next.h; validated on MI355X, tests/test_gpu_next.py); there is
no CPU / PyTorch fallback."""
from __future__ import annotations

import torch

from . import ops
from ._lib import check, lib


def letterbox_params(img1_shape, img0_shape, ratio_pad=None):
    """(gain, pad_x, pad_y) as scale_boxes derives them (utils/ops.py:141-147)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad_x = round((img1_shape[1] - round(img0_shape[1] * gain)) / 2 - 0.1)
        pad_y = round((img1_shape[0] - round(img0_shape[0] * gain)) / 2 - 0.1)
    else:
        gain = ratio_pad[0][0]
        pad_x, pad_y = ratio_pad[1]
    return gain, pad_x, pad_y


def _params(img1_shape, img0_shapes, ratio_pads, device):
    rows = []
    for i, s0 in enumerate(img0_shapes):
        gain, px, py = letterbox_params(img1_shape, s0, None if ratio_pads is None else ratio_pads[i])
        rows.append([gain, px, py, s0[1], s0[0]])
    return torch.tensor(rows, dtype=torch.float32).to(device)


def scale_detections(img1_shape, dets: torch.Tensor, counts: torch.Tensor | None, img0_shapes, ratio_pads=None, padding: bool = True,
                     xywh: bool = False) -> torch.Tensor:
    """dets fp32 [B, max_det, >=4] (as returned by nms_padded), rescaled in place image by image."""
    ops.require_gpu(dets, "yolo_master_amd post-processing")
    if dets.dtype != torch.float32 or dets.dim() != 3 or dets.stride(2) != 1 or dets.stride(0) != dets.shape[1] * dets.stride(1):
        raise ValueError("scale_detections: fp32 [B, max_det, >=4] detections with contiguous rows")
    B, max_det = dets.shape[:2]
    if len(img0_shapes) != B:
        raise ValueError("scale_detections: one original shape per image")
    params = _params(img1_shape, img0_shapes, ratio_pads, dets.device)
    check(lib.ymk_scale_boxes(ops._p(dets), dets.stride(1), ops._p(counts), ops._p(params), B, max_det, int(padding), int(xywh),
                              ops._stream()), "scale_boxes")
    return dets


def scale_boxes(img1_shape, boxes: torch.Tensor, img0_shape, ratio_pad=None, padding: bool = True, xywh: bool = False) -> torch.Tensor:
    """ultralytics.utils.ops.scale_boxes for one image: boxes fp32 [N, >=4] on the GPU, modified in place and returned."""
    if boxes.dim() != 2:
        raise ValueError("scale_boxes: boxes [N, >=4]")
    scale_detections(img1_shape, boxes.unsqueeze(0), None, [img0_shape], None if ratio_pad is None else [ratio_pad], padding, xywh)
    return boxes


def gather_mask_coefficients(mc: torch.Tensor, b: int, idx: torch.Tensor) -> torch.Tensor:
    """Mask coefficients of the anchors NMS kept for image b: mc fp32 [B, nm, A] (Segment head), idx int64 [n] (the
    `return_idxs` output of non_max_suppression) -> fp32 [n, nm]."""
    ops.require_gpu(mc, "yolo_master_amd post-processing")
    if mc.dtype != torch.float32 or not mc.is_contiguous() or idx.dtype != torch.int64 or not idx.is_contiguous():
        raise ValueError("gather_mask_coefficients: contiguous fp32 [B, nm, A] coefficients, contiguous int64 indices")
    n = idx.numel()
    out = torch.empty((n, mc.shape[1]), dtype=torch.float32, device=mc.device)
    check(lib.ymk_mask_coeff_gather(ops._p(mc), mc.shape[1], mc.shape[2], int(b), ops._p(idx), n, ops._p(out), ops._stream()), "mask_coeff_gather")
    return out


def process_mask(protos: torch.Tensor, masks_in: torch.Tensor, bboxes: torch.Tensor, shape, upsample: bool = False) -> torch.Tensor:
    """ultralytics.utils.ops.process_mask for one image (utils/ops.py:500-528), with the prototypes in the layout the Proto
    module leaves them: NHWC [mh, mw, nm] (a channel-dense view; the reference takes [nm, mh, mw]).  masks_in fp32 [n, nm],
    bboxes fp32 [n, >=4] xyxy in network-input pixels, shape = (H, W) of the network input.  Returns uint8 [n, H, W] when
    upsample else [n, mh, mw]."""
    ops.require_gpu(protos, "yolo_master_amd post-processing")
    if protos.dim() != 3 or protos.stride(2) != 1 or protos.stride(0) != protos.shape[1] * protos.stride(1):
        raise ValueError("process_mask: prototypes are an NHWC [mh, mw, nm] view of one image")
    mh, mw, nm = protos.shape
    n = masks_in.shape[0]
    H, W = (int(shape[0]), int(shape[1])) if upsample else (mh, mw)
    out = torch.empty((n, H, W), dtype=torch.uint8, device=protos.device)
    if n == 0:
        return out
    if masks_in.dtype != torch.float32 or not masks_in.is_contiguous() or tuple(masks_in.shape) != (n, nm):
        raise ValueError("process_mask: contiguous fp32 [n, nm] coefficients")
    if bboxes.dtype != torch.float32 or bboxes.dim() != 2 or bboxes.shape[0] != n or bboxes.stride(1) != 1:
        raise ValueError("process_mask: fp32 [n, >=4] boxes")
    ws = torch.empty((n * mh * mw,), dtype=torch.float32, device=protos.device)
    check(lib.ymk_process_mask(ops.DT[protos.dtype], ops._p(protos), protos.stride(1), mh, mw, nm, ops._p(masks_in), ops._p(bboxes),
                               bboxes.stride(0), n, H, W, int(bool(upsample)), float(torch.tensor(mw / shape[1], dtype=torch.float32)),
                               float(torch.tensor(mh / shape[0], dtype=torch.float32)), ops._p(ws), ops._p(out), ops._stream()), "process_mask")
    return out


# ----------------------------------------------------------------------------- validation matching (val() after NMS)
def box_iou(box1: torch.Tensor, box2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """ultralytics.utils.metrics.box_iou: fp32 [N, >=4] x [M, >=4] xyxy boxes on the GPU -> IoU [N, M]."""
    ops.require_gpu(box1, "yolo_master_amd post-processing")
    ops.require_gpu(box2, "yolo_master_amd post-processing")
    if box1.dim() != 2 or box2.dim() != 2 or box1.stride(1) != 1 or box2.stride(1) != 1 or box1.dtype != torch.float32 or box2.dtype != torch.float32:
        raise ValueError("box_iou: fp32 [N, >=4] boxes with contiguous rows")
    N, M = box1.shape[0], box2.shape[0]
    out = torch.empty((N, M), dtype=torch.float32, device=box1.device)
    check(lib.ymk_box_iou(ops._p(box1), box1.stride(0) if N else 4, N, ops._p(box2), box2.stride(0) if M else 4, M, float(eps), ops._p(out),
                          ops._stream()), "box_iou")
    return out


def match_predictions(dets: torch.Tensor, counts: torch.Tensor | None, labels: torch.Tensor, label_off: torch.Tensor, iouv: torch.Tensor,
                      eps: float = 1e-7) -> torch.Tensor:
    """BaseValidator.match_predictions (engine/validator.py:301-336) + the box_iou of DetectionValidator._process_batch for a batch.
    dets fp32 [B, max_det, >=6] (xyxy, conf, cls; nms_padded's output scaled to the labels' frame), counts int32 [B] or None,
    labels fp32 [sum L_b, 5] = (cls, x1, y1, x2, y2) grouped by image, label_off int32 [B + 1], iouv fp32 [T].
    Returns correct bool [B, max_det, T] (rows past counts[b] False)."""
    ops.require_gpu(dets, "yolo_master_amd post-processing")
    if dets.dtype != torch.float32 or dets.dim() != 3 or dets.shape[2] < 6 or not dets.is_contiguous():
        raise ValueError("match_predictions: contiguous fp32 [B, max_det, >=6] detections")
    if labels.dtype != torch.float32 or (labels.numel() and (labels.dim() != 2 or labels.shape[1] != 5 or not labels.is_contiguous())):
        raise ValueError("match_predictions: contiguous fp32 [total_labels, 5] labels (cls, x1, y1, x2, y2)")
    B, max_det = dets.shape[:2]
    T, total = int(iouv.numel()), int(labels.shape[0]) if labels.numel() else 0
    if label_off.dtype != torch.int32 or label_off.numel() != B + 1:
        raise ValueError("match_predictions: label_off int32 [B + 1]")
    nbytes = lib.ymk_match_predictions_workspace_bytes(total, T)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dets.device)
    correct = torch.empty((B, max_det, T), dtype=torch.uint8, device=dets.device)
    check(lib.ymk_match_predictions(ops._p(dets), dets.stride(1), ops._p(counts), B, max_det, ops._p(labels) if total else None,
                                    ops._p(label_off), total, ops._p(iouv.float().contiguous()), T, float(eps), ops._p(correct), ops._p(ws),
                                    nbytes, ops._stream()), "match_predictions")
    return correct.bool()


def batch_stats(dets: torch.Tensor, counts: torch.Tensor, labels: torch.Tensor, label_off: torch.Tensor, iouv: torch.Tensor) -> list[dict]:
    """The per-image statistics DetectionValidator.update_metrics hands to `metrics.update_stats`
    (models/yolo/detect/val.py:176-192, with _process_batch :313-327), for a whole batch: one matching launch and ONE
    device -> host copy instead of B x (IoU, matching, three `.cpu()` calls).  Arguments as `match_predictions`.
    Returns, per image, {"tp": bool [n, T], "conf": fp32 [n], "pred_cls": fp32 [n], "target_cls": fp32 [L], "target_img": unique
    target classes} as numpy arrays (n = counts[b]; empty predictions give `np.zeros(0)` for conf / pred_cls as the reference does)."""
    import numpy as np

    correct = match_predictions(dets, counts, labels, label_off, iouv)
    B, max_det, T = correct.shape
    # one staging buffer: [B, max_det, T + 2] = tp columns, conf, cls
    pack = torch.cat([correct.to(torch.float32), dets[:, :, 4:6]], dim=2).cpu().numpy()
    n_all = counts.cpu().numpy()
    lab = labels.cpu().numpy() if labels.numel() else np.zeros((0, 5), np.float32)
    off = label_off.cpu().numpy()
    out = []
    for b in range(B):
        n = int(n_all[b])
        cls = lab[off[b]:off[b + 1], 0]
        no_pred = n == 0
        out.append({"tp": pack[b, :n, :T] > 0.5, "conf": np.zeros(0) if no_pred else pack[b, :n, T].copy(),
                    "pred_cls": np.zeros(0) if no_pred else pack[b, :n, T + 1].copy(), "target_cls": cls.copy(), "target_img": np.unique(cls)})
    return out
