"""Segment head: the oracle's Proto / Segment restatement and the product's host path (emulated
entry points) against golden vectors from the REAL reference's SegmentationModel (tests/golden/make_golden_seg.py)."""
import json

import numpy as np
import torch

from tests.helpers import fill_by_name


def _fixture(golden_dir):
    z = np.load(golden_dir / "fwd_seg_n.npz")
    cfg, spec = json.loads(str(z["cfg"])), json.loads(str(z["spec"]))
    return z, cfg, fill_by_name(spec, seed=11, gain=0.8)


def test_segment_oracle_reproduces_reference(golden_dir):
    from oracle import model_ref
    from yolo_master_amd.nn.tasks import DetectionModel

    z, cfg, sd = _fixture(golden_dir)
    full = dict(DetectionModel(cfg).state_dict())     # supplies the fixed tensors (DFL arange, BN counters)
    full.update(sd)
    with torch.inference_mode():
        y, _, _, mc, proto = model_ref.forward(cfg, full, torch.from_numpy(z["x"]), fused=False)
    assert float((y - torch.from_numpy(z["y"])).abs().max()) <= 1e-4 * float(np.abs(z["y"]).max())
    assert float((proto - torch.from_numpy(z["proto"])).abs().max()) <= 1e-5 * max(1.0, float(np.abs(z["proto"]).max()))
    assert torch.equal(y[:, -32:], mc)


def test_segment_state_dict_contract(golden_dir):
    from yolo_master_amd.nn.tasks import DetectionModel

    keys = json.load(open(golden_dir / "keys_seg_n.json"))
    z, cfg, _ = _fixture(golden_dir)
    sd = DetectionModel(cfg).state_dict()
    assert list(sd.keys()) == list(keys.keys()) and all(list(sd[k].shape) == v for k, v in keys.items())


def test_segment_host_path_vs_reference(golden_dir, emu):
    from yolo_master_amd import ops
    from yolo_master_amd.nn.modules import Segment
    from yolo_master_amd.nn.tasks import DetectionModel

    z, cfg, sd = _fixture(golden_dir)
    m = DetectionModel(cfg)
    full = dict(m.state_dict())
    full.update(sd)
    m.load_state_dict(full)
    m.eval()
    assert isinstance(m.model[-1], Segment)
    x = torch.from_numpy(z["x"])
    with torch.inference_mode():
        y, preds = m._predict_once(x)
        ycat = torch.cat([y, preds["mask_coefficient"]], 1)
        proto = ops.nhwc_to_nchw_f32(preds["proto"])
    ref_y, ref_p = torch.from_numpy(z["y"]), torch.from_numpy(z["proto"])
    nc = m.model[-1].nc
    assert tuple(ycat.shape) == tuple(ref_y.shape) and tuple(proto.shape) == tuple(ref_p.shape)
    assert float((ycat[:, :4] - ref_y[:, :4]).abs().max()) <= 1e-3 + 1e-4 * float(ref_y[:, :4].abs().max())   # boxes, pixels
    assert float((ycat[:, 4:4 + nc] - ref_y[:, 4:4 + nc]).abs().max()) <= 1e-4                                  # scores
    assert float((ycat[:, 4 + nc:] - ref_y[:, 4 + nc:]).abs().max()) <= 1e-4 * max(1.0, float(ref_y[:, 4 + nc:].abs().max()))
    assert float((proto - ref_p).abs().max()) <= 1e-4 * max(1.0, float(ref_p.abs().max()))
    assert emu.CALLS["pixel_shuffle2"] == 1 and emu.CALLS["tokens_to_rows"] == 3
    # module-level API: the reference's eval structure ((cat(y, mc), proto), preds)
    feats = [torch.randn(1, c, s, s) for c, s in ((64, 8), (128, 4), (256, 2))]
    head = Segment(80, 32, 64, ch=(64, 128, 256)).eval()
    head.stride = torch.tensor([8.0, 16.0, 32.0])
    with torch.inference_mode():
        (yy, pp), pr = head(feats)
    assert tuple(yy.shape) == (1, 4 + 80 + 32, 64 + 16 + 4) and tuple(pp.shape) == (1, 32, 16, 16) and "mask_coefficient" in pr


def test_segmentation_model_api(golden_dir, emu):
    """`SegmentationModel("yolo-master-seg-n.yaml")` (own YAML of the cfg directory) builds the same network as the
    reference's seg YAML and returns the reference's eval structure."""
    from yolo_master_amd.nn.tasks import DetectionModel, SegmentationModel

    z, cfg, sd = _fixture(golden_dir)
    m = SegmentationModel("yolo-master-seg-n.yaml")
    keys = json.load(open(golden_dir / "keys_seg_n.json"))
    assert list(m.state_dict().keys()) == list(keys.keys())
    full = dict(m.state_dict())
    full.update(sd)
    m.load_state_dict(full)
    m.eval()
    with torch.inference_mode():
        (ycat, proto), preds = m(torch.from_numpy(z["x"]))
    ref_y, ref_p = torch.from_numpy(z["y"]), torch.from_numpy(z["proto"])
    assert float((ycat[:, 4:] - ref_y[:, 4:]).abs().max()) <= 1e-4 * max(1.0, float(ref_y[:, 4:].abs().max()))
    assert float((proto - ref_p).abs().max()) <= 1e-4 * max(1.0, float(ref_p.abs().max()))
    import pytest

    with pytest.raises(ValueError):
        SegmentationModel("yolo-master-n.yaml")
    assert isinstance(m, DetectionModel)
