"""Batched NMS front-end with the reference's signature.

``non_max_suppression`` mirrors ultralytics/utils/nms.py:13-171 (argument names, defaults, return
types: a list of ``[n_i, 6]`` tensors ``(x1, y1, x2, y2, conf, cls)`` and optionally the kept anchor
indices); the work is done for the whole batch by libymk (``ymk_nms_batched``), with a single host
sync to read the per-image counts.  ``nms_padded`` is the sync-free variant used by the benchmark
and the multi-GPU gather (fixed ``[B, max_det, 6]`` + counts).

Any number of candidates per image is handled like the reference does (utils/nms.py:142-146: the max_nms best by score reach the
greedy pass): above max_nms a device-side radix select picks them (csrc/nms.hip), e.g. the validator's conf 0.001 + multi_label on
dense scenes (up to A * nc = 672 000 candidates at 640 x 640).

Differences from the reference that are contract-level, not numerical:
  * equal scores are ordered by candidate index (the reference's ``argsort(descending=True)`` is
    unstable, its tie order is implementation-defined);
  * the wall-clock ``max_time_img`` early exit (nms.py:166-169) does not exist;
  * ``cluster=True`` enables the CW-NMS box refinement, which the reference's Python only declares
    (cfg/default.yaml:195-198); spec: examples/YOLO-Master-Cross-Platform-Edge-Deployment/cpp/src/common.cpp:150-185.
"""
from __future__ import annotations

import torch

from . import ops


def class_keep_mask(classes, nc: int, device) -> torch.Tensor:
    """`classes=[...]` (utils/nms.py:63) -> uint8 [nc] keep mask for the kernel; ids outside [0, nc) match nothing, as in
    the reference's `(x[:, 5:6] == classes).any(1)`."""
    keep = torch.zeros((nc,), dtype=torch.uint8)
    for c in (classes.tolist() if torch.is_tensor(classes) else classes):
        if float(c) == int(c) and 0 <= int(c) < nc:
            keep[int(c)] = 1
    return keep.to(device)


def nms_padded(prediction: torch.Tensor, conf_thres=0.25, iou_thres=0.45, agnostic=False, multi_label=False,
               max_det=300, max_nms=30000, max_wh=7680, cluster=False, sigma=0.1, classes=None, pack=None, nc=0, use_best=True):
    """prediction: [B, 4+nc(+extra), A] fp32 on the GPU.  Returns (dets [B,max_det,6], counts [B], idx [B,max_det],
    status [1]) without synchronising.  classes: list of class ids or a ready uint8 [nc] device mask.
    pack: optional float32 [ops.nms_pack_numel(B, max_det)] buffer the outputs are carved from (one allocation: a multi-GPU
    step gathers it with a single collective, dist.gather_packed).
    nc: number of classes when rows are carried behind the class rows (Segment: utils/nms.py:76-81); their values for the kept
    detections: ops.nms_gather_rows(prediction, nc, idx, counts).
    use_best=False: never take the producer's per-anchor best class attached to `prediction` ("the y.best contract")."""
    if prediction.dtype != torch.float32:
        prediction = prediction.float()
    prediction = prediction.contiguous()
    nc = int(nc) or prediction.shape[1] - 4
    if classes is not None and not (torch.is_tensor(classes) and classes.dtype == torch.uint8):
        classes = class_keep_mask(classes, nc, prediction.device)
    return ops.nms_batched(prediction, conf_thres, iou_thres, bool(multi_label) and nc > 1, bool(agnostic), max_det,
                           max_nms, float(max_wh), cw_sigma=float(sigma) if cluster else None, class_keep=classes, pack=pack, nc=nc,
                           use_best=use_best)


def non_max_suppression(prediction, conf_thres: float = 0.25, iou_thres: float = 0.45, classes=None,
                        agnostic: bool = False, multi_label: bool = False, labels=(), max_det: int = 300, nc: int = 0,
                        max_time_img: float = 0.05, max_nms: int = 30000, max_wh: int = 7680, rotated: bool = False,
                        end2end: bool = False, return_idxs: bool = False, cluster: bool = False, sigma: float = 0.1, use_best: bool = True):
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if rotated or end2end or prediction.shape[-1] == 6 or labels:
        raise NotImplementedError("ymk NMS covers the detect path: no rotated/end2end/autolabel modes")
    nc = int(nc) or prediction.shape[1] - 4
    extra = prediction.shape[1] - 4 - nc        # utils/nms.py:80: rows behind the classes (mask coefficients) ride along
    if extra < 0:
        raise ValueError(f"nc = {nc} but the prediction has {prediction.shape[1] - 4} rows behind the box")
    if prediction.dtype != torch.float32 or not prediction.is_contiguous():
        prediction = prediction.float().contiguous()
    dets, counts, idx, status = nms_padded(prediction, conf_thres, iou_thres, agnostic, multi_label, max_det, max_nms,
                                           max_wh, cluster, sigma, classes, nc=nc, use_best=use_best)
    if extra:                                   # output rows are (xyxy, conf, cls, mask...) as in the reference (:117,122,127)
        dets = torch.cat([dets, ops.nms_gather_rows(prediction, nc, idx, counts)], 2)
    n = counts.tolist()  # the one host sync of the post-processing step
    out = [dets[b, : n[b]] for b in range(len(n))]
    if return_idxs:
        return out, [idx[b, : n[b]].long() for b in range(len(n))]
    return out
