"""Drop-in hooks: put libymk under the REFERENCE's own objects (`ultralytics` must be importable).

    from ultralytics import YOLO
    import yolo_master_amd
    model = YOLO("yolo-master-n.yaml")            # or a .pt checkpoint
    yolo_master_amd.enable(model)                 # once, before predict()/val() fuse the model
    model.predict(source, device=0)               # reference predictor / validator, libymk forward + NMS underneath

What `enable` does (every hook falls through to the reference's own code for CPU tensors, training mode, profiling,
`visualize` / `embed` / `augment` — the conditions under which the reference leaves its fast path too):

  1. builds this package's `DetectionModel` / `SegmentationModel` from the reference model's own YAML dict
     (`model.yaml`, the dict `parse_model` consumed, nn/tasks.py:2022-2270) and loads the reference's `state_dict()` into
     it — same keys and shapes (tests/golden/keys_*.json) — so the weights are packed from the reference's parameters;
  2. rebinds the reference model's `_predict_once` (the layer loop, nn/tasks.py:182-218): eval-mode GPU batches run the
     libymk graph walk and return what the reference's Detect returns in eval mode, `(y [B, 4+nc, A], preds dict)`
     (nn/modules/head.py:157-171).  The weights are a snapshot taken at `enable()` time (the reference folds Conv+BN in
     place when its predictor starts): call `disable()` + `enable()` again after loading other weights;
  3. patches `ultralytics.utils.nms.non_max_suppression` (called as `nms.non_max_suppression(...)` by the detect
     predictor and validators, models/yolo/detect/predict.py:54, val.py:116), `ultralytics.utils.ops.scale_boxes`
     (predict.py:122) and `DetectionValidator._process_batch` (box_iou + match_predictions of `model.val()`,
     models/yolo/detect/val.py:313-327) with the libymk versions for GPU tensors.

`disable(model)` restores everything.  The hooks are per model instance (2) and per process (3)."""
from __future__ import annotations

import copy
import sys

import torch

from . import ops
from .nms import non_max_suppression as ymk_nms

_PATCHED = {}          # name -> original callable of the process-wide patches
_STATE_ATTR = "_ymk_dropin"


def _reference_core(model):
    """The reference `BaseModel` (has `.yaml` and the layer Sequential `.model`) inside a YOLO wrapper or given directly."""
    core = model
    for _ in range(3):
        if hasattr(core, "yaml") and isinstance(getattr(core, "model", None), torch.nn.Sequential):
            return core
        core = getattr(core, "model", None)
        if core is None:
            break
    raise TypeError("enable(): expected an ultralytics YOLO object or its DetectionModel / SegmentationModel")


def _build(core):
    from .nn.tasks import DetectionModel, SegmentationModel

    cfg = copy.deepcopy(core.yaml)
    head = type(core.model[-1]).__name__
    if head not in ("Detect", "Segment"):
        raise NotImplementedError(f"enable(): head {head} is not on the libymk path (Detect / Segment)")
    sd = core.state_dict()
    if any(k.endswith(".conv.bias") for k in sd):   # Conv.conv has no bias until fuse_conv_and_bn gives it one (torch_utils.py:315-349)
        raise RuntimeError("enable(): the reference model is already fused (Conv+BN folded); call enable() before predict()/val()/fuse()")
    ymk = (SegmentationModel if head == "Segment" else DetectionModel)(cfg, ch=cfg.get("channels", cfg.get("ch", 3)), nc=cfg.get("nc"))
    ymk.load_state_dict({k: v.detach().float().cpu() for k, v in sd.items()})
    ymk.stride = core.stride.clone() if torch.is_tensor(getattr(core, "stride", None)) else ymk.stride
    if hasattr(ymk.model[-1], "stride") and torch.is_tensor(getattr(core.model[-1], "stride", None)):
        ymk.model[-1].stride = core.model[-1].stride.detach().float().cpu().clone()
    return ymk.eval()


def enable(model, dtype: torch.dtype | None = None, patch_nms: bool = True):
    """Hook libymk under a reference model (see module docstring).  dtype: compute type of the libymk path (default: bf16
    when the reference model runs in half precision, fp32 otherwise).  Returns `model`."""
    core = _reference_core(model)
    if getattr(core, _STATE_ATTR, None) is not None:
        return model
    ymk = _build(core)
    state = {"ymk": ymk, "orig": core._predict_once, "dtype": dtype, "device": None, "calls": 0, "fallbacks": 0}
    # the reference fuses Conv+BN when its predictor / validator starts (nn/autobackend.py, engine/validator.py): the libymk
    # model holds its own packed (folded) copy taken here, from the unfused parameters

    def _predict_once(x, profile=False, visualize=False, embed=None):
        if core.training or profile or visualize or embed or not torch.is_tensor(x) or not ops.device_ok(x):
            state["fallbacks"] += 1
            return state["orig"](x, profile, visualize, embed)
        if state["device"] != x.device:
            ymk.to(x.device)
            state["device"] = x.device
        want = state["dtype"] or (torch.bfloat16 if x.dtype in (torch.float16, torch.bfloat16) else torch.float32)
        if getattr(ymk, "_compute_dtype", None) != want:
            ymk.set_compute_dtype(want)
            ymk._compute_dtype = want
        state["calls"] += 1
        y, preds = ymk._predict_once(x.float())
        ymk.check_flags()
        if x.dtype == torch.float16:
            y = y.half()
        return y, preds

    core._predict_once = _predict_once
    setattr(core, _STATE_ATTR, state)
    if patch_nms:
        _patch_process()
    return model


def disable(model):
    core = _reference_core(model)
    state = getattr(core, _STATE_ATTR, None)
    if state is not None:
        try:
            del core._predict_once          # the instance attribute shadows the class method
        except AttributeError:
            pass
        setattr(core, _STATE_ATTR, None)
    _unpatch_process()
    return model


def stats(model) -> dict:
    """How often the hooks ran (tests / diagnostics)."""
    s = getattr(_reference_core(model), _STATE_ATTR, None) or {}
    return {"calls": s.get("calls", 0), "fallbacks": s.get("fallbacks", 0), "nms_calls": _PATCHED.get("_nms_calls", 0),
            "match_calls": _PATCHED.get("_match_calls", 0)}


def _patch_process():
    if "nms" in _PATCHED:
        return
    import ultralytics.utils.nms as ref_nms
    import ultralytics.utils.ops as ref_ops

    from . import postprocess

    orig_nms, orig_scale = ref_nms.non_max_suppression, ref_ops.scale_boxes
    _PATCHED.update(nms=orig_nms, scale=orig_scale, _nms_calls=0)

    def non_max_suppression(prediction, *args, **kw):
        p = prediction[0] if isinstance(prediction, (list, tuple)) else prediction
        unsupported = kw.get("rotated") or kw.get("end2end") or kw.get("labels") or p.shape[-1] == 6 or \
            (kw.get("nc") and kw["nc"] != p.shape[1] - 4)
        if unsupported or not ops.device_ok(p):
            return orig_nms(prediction, *args, **kw)
        _PATCHED["_nms_calls"] += 1
        kw.pop("max_time_img", None)
        return ymk_nms(p.float(), *args, **kw)

    def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True, xywh=False):
        if torch.is_tensor(boxes) and boxes.dim() == 2 and boxes.dtype == torch.float32 and boxes.is_contiguous() and \
                boxes.is_cuda and boxes.shape[0] > 0:
            return postprocess.scale_boxes(img1_shape, boxes, img0_shape, ratio_pad, padding, xywh)
        return orig_scale(img1_shape, boxes, img0_shape, ratio_pad, padding, xywh)

    ref_nms.non_max_suppression = non_max_suppression
    ref_ops.scale_boxes = scale_boxes
    try:
        from ultralytics.models.yolo.detect.val import DetectionValidator
    except Exception:      # a trimmed reference install without the validators: the predictor hooks above still apply
        return
    orig_pb = DetectionValidator._process_batch
    _PATCHED.update(val_cls=DetectionValidator, process_batch=orig_pb, _match_calls=0)

    def _process_batch(self, preds, batch):
        pb, pc, tb, tc = preds["bboxes"], preds["cls"], batch["bboxes"], batch["cls"]
        if tc.shape[0] == 0 or pc.shape[0] == 0 or not (ops.device_ok(pb) and ops.device_ok(tb)):
            return orig_pb(self, preds, batch)
        _PATCHED["_match_calls"] += 1
        dets = torch.cat([pb.float(), preds["conf"].float().view(-1, 1), pc.float().view(-1, 1)], 1).unsqueeze(0).contiguous()
        labels = torch.cat([tc.float().view(-1, 1), tb.float()], 1).contiguous()
        off = torch.tensor([0, labels.shape[0]], dtype=torch.int32, device=dets.device)
        correct = postprocess.match_predictions(dets, None, labels, off, self.iouv.to(dets.device))
        return {"tp": correct[0].cpu().numpy()}

    DetectionValidator._process_batch = _process_batch


def _unpatch_process():
    if "nms" not in _PATCHED:
        return
    ref_nms, ref_ops = sys.modules["ultralytics.utils.nms"], sys.modules["ultralytics.utils.ops"]
    ref_nms.non_max_suppression, ref_ops.scale_boxes = _PATCHED.pop("nms"), _PATCHED.pop("scale")
    _PATCHED.pop("_nms_calls", None)
    if "val_cls" in _PATCHED:
        _PATCHED.pop("val_cls")._process_batch = _PATCHED.pop("process_batch")
        _PATCHED.pop("_match_calls", None)
