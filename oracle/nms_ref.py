"""ORACLE (test infrastructure): numpy restatement of the reference's NMS pipeline.

* non_max_suppression  — ultralytics/utils/nms.py:13-171 (candidate filter, best-class / multi-label,
  max_nms cap, class offset, greedy NMS, max_det cap), float32 arithmetic op-for-op.
* nms_greedy           — TorchNMS.nms, ultralytics/utils/nms.py:245-302.
* cw_refine            — Cluster-Weighted box refinement; the reference's Python never implements it
  (cfg/default.yaml:195-198 are dead keys); the executable spec is the C++ edge demo
  examples/YOLO-Master-Cross-Platform-Edge-Deployment/cpp/src/common.cpp:150-185 (fp64).  Pinned against that C++
  code itself, compiled in place by oracle/cwref/build.py (tests/golden/make_golden_cw.py, tests/test_oracle_cw.py).

Ordering: the reference sorts with torch ``argsort(descending=True)`` (unstable); for equal scores
this restatement (and the HIP kernel) use the lower candidate index first.  tests/golden pins that
the real reference produces the same order on the committed fixtures.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def xywh2xyxy(x):
    """ultralytics/utils/ops.py:248-264."""
    y = np.empty_like(x)
    wh = x[..., 2:] / f32(2)
    y[..., :2] = x[..., :2] - wh
    y[..., 2:] = x[..., :2] + wh
    return y


def nms_greedy(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """TorchNMS.nms (nms.py:245-302): keep i, drop j with !(IoU <= thr); returns indices, score-desc."""
    if boxes.size == 0:
        return np.zeros((0,), np.int64)
    boxes = boxes.astype(f32)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")
    keep = []
    thr = f32(thr)
    while order.size > 0:
        i = order[0]
        keep.append(i)
        if order.size == 1:
            break
        rest = order[1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(xx2 - xx1, f32(0)); h = np.maximum(yy2 - yy1, f32(0))
        inter = w * h
        if inter.sum() == 0:
            order = rest
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (areas[i] + areas[rest] - inter)
        order = rest[iou <= thr]
    return np.asarray(keep, np.int64)


def non_max_suppression(pred: np.ndarray, conf_thres=0.25, iou_thres=0.45, multi_label=False, agnostic=False,
                        max_det=300, max_nms=30000, max_wh=7680, return_idxs=False, classes=None, nc=0):
    """pred: [B, 4+nc(+extra), A] float32 (xywh + class scores + carried rows).  Returns list of [n,6+extra] (xyxy, conf, cls, extra).
    classes: keep only candidates of these class ids (utils/nms.py:63,131-136: after the best-class choice).
    nc: number of classes (utils/nms.py:76-81: `nc = nc or shape[1] - 4`, `extra = shape[1] - nc - 4` mask columns are carried)."""
    pred = np.asarray(pred, f32)
    B, ch, A = pred.shape
    nc = nc or ch - 4
    extra = ch - nc - 4
    mi = 4 + nc
    multi_label = multi_label and nc > 1
    conf_thres = f32(conf_thres)
    xc = pred[:, 4:mi].max(1) > conf_thres
    p = np.transpose(pred, (0, 2, 1)).copy()
    p[..., :4] = xywh2xyxy(p[..., :4])
    outs, idxs = [], []
    for b in range(B):
        x = p[b][xc[b]]
        xk = np.arange(A)[xc[b]]
        if x.shape[0] == 0:
            outs.append(np.zeros((0, 6 + extra), f32)); idxs.append(np.zeros((0,), np.int64)); continue
        box, cls, mask = x[:, :4], x[:, 4:mi], x[:, mi:]
        if multi_label:
            i, j = np.where(cls > conf_thres)
            x = np.concatenate((box[i], cls[i, j][:, None], j[:, None].astype(f32), mask[i]), 1)
            xk = xk[i]
        else:
            j = cls.argmax(1)
            conf = cls[np.arange(cls.shape[0]), j]
            filt = conf > conf_thres
            x = np.concatenate((box, conf[:, None], j[:, None].astype(f32), mask), 1)[filt]
            xk = xk[filt]
        if classes is not None:
            filt = (x[:, 5:6] == np.asarray(classes, f32).reshape(1, -1)).any(1)
            x, xk = x[filt], xk[filt]
        n = x.shape[0]
        if n == 0:
            outs.append(np.zeros((0, 6 + extra), f32)); idxs.append(np.zeros((0,), np.int64)); continue
        if n > max_nms:
            filt = np.argsort(-x[:, 4], kind="stable")[:max_nms]
            x, xk = x[filt], xk[filt]
        c = x[:, 5:6] * f32(0 if agnostic else max_wh)
        boxes = x[:, :4] + c
        keep = nms_greedy(boxes, x[:, 4], iou_thres)[:max_det]
        outs.append(x[keep]); idxs.append(xk[keep])
    return (outs, idxs) if return_idxs else outs


def cw_refine(cands: np.ndarray, keep: np.ndarray, iou_thres: float, sigma: float, pool_cap: int = 3000, agnostic: bool = False):
    """CW-NMS refinement per common.cpp:150-185 (fp64).  cands: [n,6] (xyxy, conf, cls) — the
    conf-filtered candidates of one image; keep: indices of the greedy survivors.  Returns [len(keep),4].
    agnostic (not in the C++ spec, which only has per-class NMS): clusters ignore the class, like the suppression did."""
    c = cands.astype(np.float64)
    order = np.argsort(-cands[:, 4], kind="stable")[:pool_cap]
    out = np.zeros((len(keep), 4))
    for s, k in enumerate(keep):
        kb, kc = c[k, :4], c[k, 5]
        ak = (kb[2] - kb[0]) * (kb[3] - kb[1])
        sw = 0.0
        acc = np.zeros(4)
        for m in order:
            if not agnostic and c[m, 5] != kc:
                continue
            mb = c[m, :4]
            iw = min(kb[2], mb[2]) - max(kb[0], mb[0]); ih = min(kb[3], mb[3]) - max(kb[1], mb[1])
            if iw <= 0 or ih <= 0:
                continue
            inter = iw * ih
            uni = ak + (mb[2] - mb[0]) * (mb[3] - mb[1]) - inter
            ov = inter / uni if uni > 0 else 0.0
            if ov <= iou_thres:
                continue
            w = c[m, 4] * np.exp(-((1.0 - ov) ** 2) / sigma)
            sw += w
            acc += w * mb
        if sw > 1e-6:
            x0, y0 = acc[0] / sw, acc[1] / sw
            out[s] = [x0, y0, x0 + max(0.0, acc[2] / sw - x0), y0 + max(0.0, acc[3] / sw - y0)]
        else:
            out[s] = kb
    return out
