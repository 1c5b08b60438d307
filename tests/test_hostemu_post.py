"""ymk_scale_boxes (include/ymk_next.h, csrc/post.hip) on the CPU lane emulator through the product's own wrapper
(yolo_master_amd/postprocess.py): bit-exact against the REAL reference's golden vectors, single image and batched."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.hostemu import build as hostemu_build
from tests.test_oracle_post import cases


@pytest.fixture
def post(monkeypatch):
    path = hostemu_build.build()
    if path is None:
        pytest.skip("no host clang++ to build the kernel emulation")
    from yolo_master_amd import _lib, ops, postprocess

    h = C.CDLL(str(path))
    h.ymk_scale_boxes.restype, h.ymk_scale_boxes.argtypes = _lib.SYMBOLS_NEXT["ymk_scale_boxes"]
    monkeypatch.setenv("YMK_EXPERIMENTAL", "1")
    monkeypatch.setattr(postprocess, "lib", h)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    return postprocess


def test_scale_boxes_single_image(post, golden_dir):
    for c in cases(golden_dir):
        boxes = torch.from_numpy(c["boxes"].copy())                       # [N, 6]: the kernel touches the first four columns only
        out = post.scale_boxes(c["img1"], boxes, c["img0"], ratio_pad=c["ratio_pad"], padding=c["padding"], xywh=c["xywh"])
        assert out is boxes
        assert np.array_equal(boxes[:, :4].numpy(), c["out"]), (c["img1"], c["img0"])
        assert np.array_equal(boxes[:, 4:].numpy(), c["boxes"][:, 4:])


def test_scale_detections_batched(post, golden_dir):
    cs = [c for c in cases(golden_dir) if c["img1"] == (640, 640) and c["padding"] and not c["xywh"] and c["ratio_pad"] is None]
    assert len(cs) >= 4
    max_det = 50
    dets = torch.full((len(cs), max_det, 6), -3.0)
    counts = torch.tensor([37, 20, 0, 37, 5][: len(cs)], dtype=torch.int32)
    for b, c in enumerate(cs):
        dets[b, :37] = torch.from_numpy(c["boxes"])
    before = dets.clone()
    post.scale_detections((640, 640), dets, counts, [c["img0"] for c in cs])
    for b, c in enumerate(cs):
        n = int(counts[b])
        assert np.array_equal(dets[b, :n, :4].numpy(), c["out"][:n])
        assert torch.equal(dets[b, n:], before[b, n:]) and torch.equal(dets[b, :, 4:], before[b, :, 4:])   # nothing else touched


def test_scale_boxes_is_opt_in(monkeypatch):
    from yolo_master_amd import ops, postprocess

    monkeypatch.delenv("YMK_EXPERIMENTAL", raising=False)
    with pytest.raises(ops.KernelNotBuilt):
        postprocess.scale_boxes((640, 640), torch.zeros(3, 4), (480, 640))
