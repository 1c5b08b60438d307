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
     place when its predictor starts): call `disable()` + `enable()` again after loading other weights.  The hook is a callable
     bound to the model that reads its state from it: a `copy.deepcopy` of the model (Exporter, ModelEMA) runs the REFERENCE path until
     `enable()` is called on the copy, and tracing / ONNX export always do;
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


def _compute_dtype_for(x: torch.Tensor, forced):
    """Compute type of the libymk path for an input tensor: the reference's reduced-precision mode is fp16 (`half=True`,
    engine/predictor.py:174,415; nn/backends/pytorch.py:67) and maps to libymk's fp16 kernels; bf16 tensors to bf16; anything else fp32."""
    if forced is not None:
        return forced
    if x.dtype == torch.float16:
        return torch.float16 if ops.HAS_F16 else torch.bfloat16
    return torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32


def _hooked_predict_once(self, x, profile=False, visualize=False, embed=None):
    """Replacement for `BaseModel._predict_once` (nn/tasks.py:182-218), installed as a callable bound to the model (`_Hook`): all state is read from `self`,
    so a `copy.deepcopy` of the model (the reference's Exporter, ModelEMA) gets a hook that looks at the COPY — its training flag,
    its own state entry — and the copy's state is detached (below) so that it runs the reference path instead of a stale weight
    snapshot.  Tracing / ONNX export always take the reference path (libymk kernels are opaque to the tracer)."""
    state = self.__dict__.get(_STATE_ATTR)
    orig = type(self)._predict_once
    tracing = torch.jit.is_tracing() or torch.jit.is_scripting() or (hasattr(torch.onnx, "is_in_onnx_export") and torch.onnx.is_in_onnx_export())
    # (`inert`: a deep copy's marker — an id can be reused once the original is freed, so the owner id alone does not prove liveness)
    live = state is not None and state.get("owner") == id(self) and not state.get("inert") and "ymk" in state
    if not live or self.training or tracing or profile or visualize or embed or not torch.is_tensor(x) or not ops.device_ok(x):
        if live:
            state["fallbacks"] += 1
        return orig(self, x, profile, visualize, embed)
    ymk = state["ymk"]
    if state["device"] != x.device:
        ymk.to(x.device)
        state["device"] = x.device
    want = _compute_dtype_for(x, state["dtype"])
    if getattr(ymk, "_compute_dtype", None) != want:
        ymk.set_compute_dtype(want)
        ymk._compute_dtype = want
    state["calls"] += 1
    y, preds = ymk._predict_once(x.float())
    ymk.check_flags()
    low = x.dtype in (torch.float16, torch.bfloat16)
    if "mask_coefficient" in preds:      # Segment in eval mode returns ((cat(y, mask coefficients), prototypes NCHW), preds) (nn/modules/head.py:317-336)
        proto = ops.nhwc_to_nchw_f32(preds["proto"])
        y = torch.cat([y, preds["mask_coefficient"]], 1)
        preds["proto"] = proto
        return ((y.to(x.dtype), proto.to(x.dtype)) if low else (y, proto)), preds
    if low:
        y = y.to(x.dtype)
    return y, preds


class _State(dict):
    """Per-model hook state.  Deep copies of the model share nothing with it: the copy's entry is an inert marker (owner id of the
    ORIGINAL), so the copy's bound hook falls through to the reference path; `enable(copy)` installs a fresh state.  Pickled
    (`torch.save(model)`, the reference's `Model.save` / trainer checkpoints) it is an empty plain dict: a checkpoint never
    references this package."""

    def __deepcopy__(self, memo):
        return _State(owner=self.get("owner"), inert=True)

    def __reduce__(self):
        return (dict, ())


class _Hook:
    """The instance attribute that shadows `BaseModel._predict_once` on an enabled model: calls `_hooked_predict_once(core, ...)`.
    A callable object rather than a bound method because a bound method pickles as `getattr(core, "_hooked_predict_once")`, which
    cannot be resolved at load time: an enabled model (or the deepcopy the reference's `Model.save` makes of it) would write a
    checkpoint that fails to load.  Pickled, the hook is the reference's OWN method bound to the model (the checkpoint loads without
    this package and runs the reference path); deep-copied, it is a hook on the copy (which falls through until `enable(copy)`)."""

    __slots__ = ("core",)

    def __init__(self, core):
        self.core = core

    def __call__(self, x, profile=False, visualize=False, embed=None):
        return _hooked_predict_once(self.core, x, profile, visualize, embed)

    def __deepcopy__(self, memo):
        return _Hook(copy.deepcopy(self.core, memo))

    def __reduce__(self):
        # getattr(core, "_predict_once") at load time: the hook sits in the model's state, so the model object exists (and is in
        # the pickle memo) but its __dict__ is not restored yet — the lookup finds the CLASS's method and binds it to the model
        return (getattr, (self.core, "_predict_once"))


def enable(model, dtype: torch.dtype | None = None, patch_nms: bool = True):
    """Hook libymk under a reference model (see module docstring).  dtype: compute type of the libymk path (default: what the
    input tensor asks for — fp16 for the reference's `half=True`, bf16 for bf16 tensors, fp32 otherwise).  Returns `model`."""
    core = _reference_core(model)
    st = core.__dict__.get(_STATE_ATTR)
    if st is not None and st.get("owner") == id(core) and not st.get("inert"):
        return model
    ymk = _build(core)
    # the reference fuses Conv+BN when its predictor / validator starts (nn/autobackend.py, engine/validator.py): the libymk
    # model holds its own packed (folded) copy taken here, from the unfused parameters
    state = _State(ymk=ymk, dtype=dtype, device=None, calls=0, fallbacks=0, owner=id(core), patched=bool(patch_nms))
    core.__dict__[_STATE_ATTR] = state
    core.__dict__["_predict_once"] = _Hook(core)   # deepcopy -> a hook on the copy; pickle -> the reference's own bound method
    if patch_nms:
        _PATCHED["_users"] = _PATCHED.get("_users", 0) + 1
        _patch_process()
    return model


def disable(model):
    core = _reference_core(model)
    state = core.__dict__.get(_STATE_ATTR)
    if state is not None:
        core.__dict__.pop("_predict_once", None)    # the instance attribute shadows the class method
        core.__dict__.pop(_STATE_ATTR, None)
        if state.get("patched") and state.get("owner") == id(core) and not state.get("inert"):
            _PATCHED["_users"] = max(_PATCHED.get("_users", 1) - 1, 0)
            if _PATCHED["_users"] == 0:             # other enabled models still rely on the process-wide patches
                _unpatch_process()
    return model


def stats(model) -> dict:
    """How often the hooks ran (tests / diagnostics)."""
    s = _reference_core(model).__dict__.get(_STATE_ATTR) or {}
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

    import inspect

    sig = inspect.signature(orig_nms)
    ymk_params = set(inspect.signature(ymk_nms).parameters)

    def non_max_suppression(prediction, *args, **kw):
        """Arguments are bound with the reference's own signature, so modes passed positionally are seen too; anything outside the
        detect / segment path (rotated, end2end, autolabels), CPU tensors and arguments this NMS does not know go to the
        reference implementation — as does a call that raises NotImplementedError (counted as a fallback, never a crash)."""
        try:
            ba = sig.bind(prediction, *args, **kw)
        except TypeError:
            return orig_nms(prediction, *args, **kw)
        a = dict(ba.arguments)
        pred = a.pop(next(iter(sig.parameters)))
        p = pred[0] if isinstance(pred, (list, tuple)) else pred
        unsupported = a.get("rotated") or a.get("end2end") or (a.get("labels") is not None and len(a.get("labels")) > 0) or \
            not torch.is_tensor(p) or p.dim() != 3 or p.shape[-1] == 6 or (a.get("nc") and a["nc"] > p.shape[1] - 4) or \
            any(k not in ymk_params for k in a if k != "max_time_img")
        if unsupported or not ops.device_ok(p):
            return orig_nms(prediction, *args, **kw)
        a.pop("max_time_img", None)
        try:
            out = ymk_nms(p.float(), **a)
        except NotImplementedError:
            _PATCHED["_nms_fallbacks"] = _PATCHED.get("_nms_fallbacks", 0) + 1
            return orig_nms(prediction, *args, **kw)
        _PATCHED["_nms_calls"] += 1
        return out

    def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True, xywh=False):
        # the predictor passes `pred[:, :4]`, a row-stride-6 view (models/yolo/detect/predict.py:122): rows need not be dense
        if torch.is_tensor(boxes) and boxes.dim() == 2 and boxes.dtype == torch.float32 and boxes.shape[1] == 4 and boxes.stride(1) == 1 \
                and ops.device_ok(boxes) and boxes.shape[0] > 0:
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
    _PATCHED.pop("_nms_fallbacks", None)
    if "val_cls" in _PATCHED:
        _PATCHED.pop("val_cls")._process_batch = _PATCHED.pop("process_batch")
        _PATCHED.pop("_match_calls", None)


# ----------------------------------------------------------------------------------------------- the third surface: an inference backend
def backend_class():
    """`YmkBackend`, a subclass of the reference's `PyTorchBackend` (`ultralytics/nn/backends/pytorch.py:16`, base class
    `nn/backends/base.py:41-140`), built on first use because the base class lives in the reference package.

    `load_model` does what the reference's does — accept an `nn.Module` or a `.pt` path, fuse, `.half()` / `.float()`, freeze — with
    one step in front: libymk is hooked under the model (`enable`) while its Conv + BatchNorm pairs are still separate, because the
    packed copy is folded from the unfused parameters.  `forward` is inherited: `self.model(im, augment=, visualize=, embed=)` reaches the
    hooked `_predict_once` for plain eval batches on the GPU and the reference's own graph walk for everything else.  `fp16=True`
    selects the fp16 build of the library (the reference's `half=True`).  Attributes (`stride`, `names`, `channels`, `end2end`,
    `kpt_shape`) are set by the inherited code from the reference model, as for the PyTorch backend."""
    if "backend_cls" in _PATCHED:
        return _PATCHED["backend_cls"]
    from ultralytics.nn.backends.pytorch import PyTorchBackend

    class YmkBackend(PyTorchBackend):
        def load_model(self, weight):
            if not isinstance(weight, torch.nn.Module):
                from ultralytics.nn.tasks import load_checkpoint

                weight, _ = load_checkpoint(weight, device=self.device, fuse=False)
            self.ymk_enabled = False
            try:
                enable(weight, dtype=torch.float16 if (self.fp16 and ops.HAS_F16) else None)
                self.ymk_enabled = True
            except (KeyError, ValueError, TypeError, NotImplementedError, RuntimeError) as e:   # a model this package does not build (or an already fused one): plain PyTorch backend
                self.ymk_error = f"{type(e).__name__}: {e}"
            super().load_model(weight)

        def stats(self):
            return stats(self.model)

    _PATCHED["backend_cls"] = YmkBackend
    return YmkBackend


def register_backend():
    """Make the reference's `AutoBackend` (`nn/autobackend.py:143-222`) construct `YmkBackend` for format "pt" (an `nn.Module` or a
    `.pt` checkpoint): `predict()` / `val()` / `AutoBackend(model, device, fp16=...)` then run libymk without an `enable()` call.
    Returns the class.  `unregister_backend()` restores the map."""
    from ultralytics.nn.autobackend import AutoBackend

    cls = backend_class()
    if "backend_prev" not in _PATCHED:
        _PATCHED["backend_prev"] = AutoBackend._BACKEND_MAP["pt"]
    AutoBackend._BACKEND_MAP["pt"] = cls
    return cls


def unregister_backend():
    if "backend_prev" in _PATCHED:
        from ultralytics.nn.autobackend import AutoBackend

        AutoBackend._BACKEND_MAP["pt"] = _PATCHED.pop("backend_prev")

