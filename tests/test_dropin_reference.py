"""BASELINE config 1 (plumbing): the REFERENCE's own `YOLO(yaml).predict(tensor)` with libymk hooked underneath by
`yolo_master_amd.enable(model)` — module registry / layer loop / NMS hook / Results construction all the reference's code.

Runs in the build container (needs /root/reference; the GPU box has no checkout).  No GPU here, so libymk's entry points are
the torch restatement of the C-ABI contract (tests/emu_ops.py, the `emu` fixture): what is tested is the HOOKS — that the
reference predictor walks through this package's graph, that the weights were taken from the reference's parameters, that
the detections equal the un-hooked reference's — not kernel numerics (those are the -m gpu tests)."""
import pytest
import torch

from oracle import refboot

pytestmark = pytest.mark.skipif(not refboot.available(), reason="reference checkout not present")
YAML = f"{refboot.REF}/ultralytics/cfg/models/master/v0/det/yolo-master-n.yaml"


def _yolo():
    refboot.boot()
    refboot.stub_torchvision()
    from ultralytics import YOLO

    from yolo_master_amd.weights import synth_state_dict

    m = YOLO(YAML, verbose=False)
    m.model.load_state_dict(synth_state_dict(m.model.state_dict(), seed=0))
    return m


def test_predict_through_the_hooks(emu, hostlib, monkeypatch):
    import yolo_master_amd
    from yolo_master_amd import dropin, ops, postprocess
    from yolo_master_amd.weights import synth_input

    monkeypatch.setattr(ops, "device_ok", lambda t: True)     # CPU tensors go to the (emulated) libymk path
    monkeypatch.setattr(postprocess, "lib", hostlib)          # scale_boxes hook: csrc/post.hip compiled for the host (lane emulator)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    scale_calls = []
    real_scale = postprocess.scale_boxes
    monkeypatch.setattr(postprocess, "scale_boxes", lambda *a, **k: (scale_calls.append(1), real_scale(*a, **k))[1])
    x = synth_input(2, 128, 128, seed=42)
    kw = dict(conf=0.002, iou=0.7, verbose=False, device="cpu")
    ref = _yolo().predict(x, **kw)                            # the untouched reference
    m = _yolo()
    assert yolo_master_amd.enable(m) is m
    try:
        got = m.predict(x, **kw)
        st = dropin.stats(m)
        assert st["calls"] >= 1 and st["nms_calls"] >= 1, st          # the batch (the reference skips warm-up on cpu); NMS hook used
        assert emu.CALLS["conv2d_stem"] >= 1 and emu.CALLS["esmoe_route"] >= 4 and emu.CALLS["nms_batched"] >= 1
        assert len(scale_calls) >= 2, "the predictor's scale_boxes(pred[:, :4]) (a row-stride-6 view) must reach libymk's kernel"
        assert len(got) == len(ref) == 2
        n = 0
        for g, r in zip(got, ref):
            gb, rb = g.boxes.data, r.boxes.data
            assert gb.shape == rb.shape and gb.shape[0] > 0
            assert torch.equal(gb[:, 5], rb[:, 5]), "classes differ"
            assert float((gb[:, 4] - rb[:, 4]).abs().max()) <= 1e-4
            assert float((gb[:, :4] - rb[:, :4]).abs().max()) <= 1e-2
            assert g.orig_shape == r.orig_shape and g.names == r.names
            n += gb.shape[0]
        print(f"reference predict() through the libymk hooks: {n} detections identical in class, <= 1e-4 in score")
        # training-mode / profiling calls are left to the reference
        core = m.model
        before = dropin.stats(m)["fallbacks"]
        core._predict_once(x, profile=False, visualize=False, embed=[1])
        assert dropin.stats(m)["fallbacks"] == before + 1
    finally:
        yolo_master_amd.disable(m)
    import ultralytics.utils.nms as ref_nms

    assert ref_nms.non_max_suppression.__module__ == "ultralytics.utils.nms", "disable() must restore the reference's NMS"
    assert "_predict_once" not in m.model.__dict__


def test_enable_refuses_a_fused_model(emu):
    import yolo_master_amd

    m = _yolo()
    m.model.fuse(verbose=False)
    with pytest.raises(RuntimeError, match="already fused"):
        yolo_master_amd.enable(m)


def test_val_style_nms_call_goes_through_the_hook(emu, monkeypatch):
    """The validator's call shape (models/yolo/detect/val.py:116: multi_label, agnostic, max_det by keyword, nc=0)."""
    import yolo_master_amd
    from yolo_master_amd import dropin, ops
    from yolo_master_amd.weights import synth_input

    monkeypatch.setattr(ops, "device_ok", lambda t: True)
    m = _yolo()
    yolo_master_amd.enable(m)
    try:
        import ultralytics.utils.nms as ref_nms

        with torch.inference_mode():
            y, _ = m.model.eval()(synth_input(1, 64, 64, seed=3))
        out = ref_nms.non_max_suppression(y, 0.05, 0.6, nc=0, multi_label=True, agnostic=False, max_det=100, end2end=False, rotated=False)
        want = dropin._PATCHED["nms"](y.clone(), 0.05, 0.6, nc=0, multi_label=True, agnostic=False, max_det=100, max_time_img=10.0)
        assert dropin.stats(m)["nms_calls"] >= 1
        assert len(out) == 1 and out[0].shape == want[0].shape and torch.allclose(out[0], want[0], atol=1e-5)
    finally:
        yolo_master_amd.disable(m)


def test_validator_matching_goes_through_the_hook(hostlib, monkeypatch, golden_dir):
    """`model.val()`'s per-image matching (DetectionValidator._process_batch, models/yolo/detect/val.py:313-327) under the hook:
    the reference validator object, the host-compiled ymk_match_predictions underneath, equal to the reference's own result on
    the real-reference fixture cases."""
    import numpy as np

    import yolo_master_amd
    from tests.test_oracle_post import match_cases
    from yolo_master_amd import dropin, ops, postprocess

    monkeypatch.setattr(postprocess, "lib", hostlib)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    monkeypatch.setattr(ops, "device_ok", lambda t: True)
    m = _yolo()
    yolo_master_amd.enable(m)
    try:
        from ultralytics.models.yolo.detect.val import DetectionValidator

        v = DetectionValidator.__new__(DetectionValidator)
        v.iouv = torch.linspace(0.5, 0.95, 10)
        v.niou = 10
        n = 0
        for c in match_cases(golden_dir):
            if c["tied"]:
                continue
            d, l = torch.from_numpy(c["dets"].copy()), torch.from_numpy(c["labels"].copy())
            preds = {"bboxes": d[:, :4], "conf": d[:, 4], "cls": d[:, 5]}
            batch = {"bboxes": l[:, 1:], "cls": l[:, 0]}
            got = v._process_batch(preds, batch)["tp"]
            want = dropin._PATCHED["process_batch"](v, preds, batch)["tp"]
            assert got.dtype == want.dtype == bool and got.shape == want.shape and np.array_equal(got, want)
            n += 1
        assert n >= 4 and dropin.stats(m)["match_calls"] >= 3
    finally:
        yolo_master_amd.disable(m)
    from ultralytics.models.yolo.detect.val import DetectionValidator as V2

    assert V2._process_batch.__module__ == "ultralytics.models.yolo.detect.val", "disable() must restore the validator"


def test_copies_traces_and_foreign_nms_modes_take_the_reference_path(emu, monkeypatch):
    """Hook hygiene (round-2 advisor findings): (1) a `copy.deepcopy` of an enabled model — what the reference's Exporter and
    ModelEMA make — must not run the ORIGINAL's weight snapshot nor look at the original's training flag: its hook is bound to the
    copy and falls through to the reference until `enable(copy)`; (2) a second enabled model keeps the process-wide patches alive
    when the first is disabled; (3) NMS modes outside the detect path are recognised when passed POSITIONALLY and go to the
    reference's implementation instead of raising."""
    import copy

    import yolo_master_amd
    from yolo_master_amd import dropin, ops
    from yolo_master_amd.weights import synth_input

    monkeypatch.setattr(ops, "device_ok", lambda t: True)
    x = synth_input(1, 64, 64, seed=3)
    m, m2 = _yolo(), _yolo()
    yolo_master_amd.enable(m)
    yolo_master_amd.enable(m2)
    try:
        import ultralytics.utils.nms as ref_nms

        core = m.model.eval()
        with torch.inference_mode():
            y, _ = core(x)
        y = y.clone()                                      # a normal tensor: the reference's NMS writes into its input
        calls = dropin.stats(m)["calls"]
        assert calls >= 1
        dup = copy.deepcopy(core)
        assert dup._predict_once.core is dup, "the hook of a copy must be bound to the copy"
        # (1b) round-3 advisor finding: an enabled model (and the deepcopy `Model.save` makes) must pickle into a checkpoint that
        # LOADS — as the plain reference model: its own `_predict_once`, no object of this package inside
        import io
        import pickle

        for obj in (core, dup):
            buf = io.BytesIO()
            torch.save(obj, buf)
            assert b"yolo_master_amd" not in buf.getvalue(), "a checkpoint must not reference the drop-in package"
            back = torch.load(io.BytesIO(buf.getvalue()), weights_only=False)
            assert back._predict_once.__func__ is type(back)._predict_once and back._predict_once.__self__ is back
            assert type(back.__dict__[dropin._STATE_ATTR]) is dict and not back.__dict__[dropin._STATE_ATTR]
            with torch.inference_mode():
                y_back, _ = back.eval()(x)                 # the reference path of the loaded model
            assert torch.allclose(y_back, y, atol=1e-3)
            assert pickle.loads(pickle.dumps(obj.__dict__["_predict_once"])).__self__ is not obj
            yolo_master_amd.enable(back)                   # ... and it can be enabled again
            with torch.inference_mode():
                back(x)
            assert dropin.stats(back)["calls"] == 1
            yolo_master_amd.disable(back)
        assert dropin.stats(m)["calls"] == calls
        with torch.inference_mode():
            y_dup, _ = dup(x)                              # reference path of the copy (its own parameters)
        assert dropin.stats(m)["calls"] == calls, "a deep copy must not run the original's libymk snapshot"
        assert torch.allclose(y_dup, y, atol=1e-3)
        dup.train()                                        # the copy's flag is the one its hook reads
        assert core.training is False
        yolo_master_amd.enable(dup.eval())                 # a copy can be enabled on its own
        with torch.inference_mode():
            dup(x)
        assert dropin.stats(dup)["calls"] == 1
        yolo_master_amd.disable(dup)
        # (2) refcount of the process-wide patches
        yolo_master_amd.disable(m)
        assert ref_nms.non_max_suppression.__module__ != "ultralytics.utils.nms", "m2 still relies on the NMS hook"
        # (3) positional foreign modes: classes, agnostic, multi_label, labels=[...] (autolabelling) -> reference implementation
        before = dropin.stats(m2)["nms_calls"]
        lab = [torch.zeros((0, 5))]
        out = ref_nms.non_max_suppression(y.clone(), 0.05, 0.6, None, False, False, lab)
        want = dropin._PATCHED["nms"](y.clone(), 0.05, 0.6, None, False, False, lab)
        assert dropin.stats(m2)["nms_calls"] == before and torch.equal(out[0], want[0])
        out = ref_nms.non_max_suppression(y, 0.05, 0.6, None, False, True)      # positional multi_label: on the libymk path
        assert dropin.stats(m2)["nms_calls"] == before + 1 and out[0].shape[1] == 6
    finally:
        yolo_master_amd.disable(m2)
        yolo_master_amd.disable(m)
    import ultralytics.utils.nms as ref_nms

    assert ref_nms.non_max_suppression.__module__ == "ultralytics.utils.nms"


def test_backend_adapter_under_autobackend(emu, monkeypatch):
    """The third plugin surface: `dropin.register_backend()` puts `YmkBackend` — a subclass of the reference's
    PyTorchBackend — behind format "pt" of the reference's AutoBackend.  An un-hooked reference model handed to AutoBackend then runs
    libymk (no enable() call), returns what the PyTorch backend returns, carries the same attributes, and unregister restores the map."""
    refboot.boot()
    refboot.stub_torchvision()
    from ultralytics.nn.autobackend import AutoBackend
    from ultralytics.nn.backends.pytorch import PyTorchBackend

    from yolo_master_amd import dropin, ops
    from yolo_master_amd.weights import synth_input

    monkeypatch.setattr(ops, "device_ok", lambda t: True)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    x = synth_input(2, 64, 64, seed=7)
    dev = torch.device("cpu")
    ref_backend = AutoBackend(_yolo().model, device=dev, fp16=False, fuse=True, verbose=False).eval()   # the predictor's setup_model does the same (engine/predictor.py:397-400)
    assert type(ref_backend.backend) is PyTorchBackend
    with torch.inference_mode():
        want = ref_backend(x)
    want = want[0] if isinstance(want, (list, tuple)) else want
    cls = dropin.register_backend()
    try:
        assert AutoBackend._BACKEND_MAP["pt"] is cls and issubclass(cls, PyTorchBackend)
        before = emu.CALLS.get("conv2d_stem", 0)
        ab = AutoBackend(_yolo().model, device=dev, fp16=False, fuse=True, verbose=False).eval()
        assert type(ab.backend) is cls and ab.backend.ymk_enabled
        assert (ab.backend.stride, ab.backend.channels, ab.backend.end2end) == (ref_backend.backend.stride, ref_backend.backend.channels, ref_backend.backend.end2end)
        assert ab.backend.names == ref_backend.backend.names
        with torch.inference_mode():
            got = ab(x)
        got = got[0] if isinstance(got, (list, tuple)) else got
        assert emu.CALLS.get("conv2d_stem", 0) > before and ab.backend.stats()["calls"] >= 1, "the batch did not go through libymk"
        assert got.shape == want.shape
        err = (got.float() - want.float()).abs()
        assert float(err[:, 4:].max()) <= 1e-4 and float(err[:, :4].max()) <= 1e-2
        dropin.disable(ab.backend.model)
    finally:
        dropin.unregister_backend()
    assert AutoBackend._BACKEND_MAP["pt"] is PyTorchBackend

