"""ymk_scale_boxes (include/ymk_next.h, csrc/post.hip) on the CPU lane emulator through the product's own wrapper
(yolo_master_amd/postprocess.py): bit-exact against the REAL reference's golden vectors, single image and batched."""
import numpy as np
import pytest
import torch

from tests.test_oracle_post import cases


@pytest.fixture
def post(hostlib, monkeypatch):
    """yolo_master_amd.postprocess / ops wired to the host-compiled kernels (tests/conftest.py: hostlib)."""
    from yolo_master_amd import ops, postprocess

    monkeypatch.setattr(postprocess, "lib", hostlib)
    monkeypatch.setattr(ops, "lib", hostlib)
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


def test_scale_boxes_has_no_cpu_path():
    from yolo_master_amd import postprocess

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        postprocess.scale_boxes((640, 640), torch.zeros(3, 4), (480, 640))


def seg_kernel_checks(dev="cpu"):
    """pixel_shuffle2 / tokens_to_rows against their contract restatements; shared with tests/test_gpu_next.py."""
    from tests import emu_ops
    from yolo_master_amd import ops

    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float32, torch.bfloat16):
        vec = 8 if dtype == torch.bfloat16 else 4
        for B, H, W, C, pad in ((2, 5, 7, 2 * vec, 0), (1, 1, 1, vec, vec), (3, 4, 3, 4 * vec, 2 * vec)):
            t = torch.randn(B, H, W, 4 * C, generator=g).to(dtype)
            wide = torch.zeros((B, H, W, 4 * C + pad), dtype=dtype)
            wide[..., : 4 * C] = t
            got = ops.pixel_shuffle2(wide.to(dev)[..., : 4 * C])
            assert torch.equal(got.cpu(), emu_ops.pixel_shuffle2(t)), (dtype, B, H, W, C)
        y = torch.full((2, 40, 100), -1.0)
        ref = y.clone()
        yd = y.to(dev)
        a_off = 0
        for H, W, C in ((6, 8, 32), (3, 4, 32), (2, 2, 32), (7, 3, 5)):
            x = torch.randn(2, H, W, C, generator=g).to(dtype)
            row_off = 0 if C == 32 else 33
            ops.tokens_to_rows(x.to(dev), yd, a_off if C == 32 else 0, row_off)
            emu_ops.tokens_to_rows(x, ref, a_off if C == 32 else 0, row_off)
            a_off += H * W if C == 32 else 0
        assert torch.equal(yd.cpu(), ref), dtype


def test_segment_kernels_on_the_emulator(post):
    seg_kernel_checks()


def mask_kernel_checks(golden_dir, dev="cpu"):
    """gather + process_mask through the product wrapper against the REAL reference's masks; shared with the GPU test."""
    from tests.test_oracle_post import mask_cases
    from yolo_master_amd import postprocess

    for c in mask_cases(golden_dir):
        for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 2e-2)):
            protos = c["protos"].permute(1, 2, 0).contiguous().to(dtype).to(dev)        # NHWC, as the Proto module leaves them
            got = postprocess.process_mask(protos, c["coefs"].to(dev), c["boxes"].to(dev), c["shape"], upsample=c["upsample"]).cpu()
            assert got.shape == c["mask"].shape and got.dtype == torch.uint8
            if got.numel():
                assert float((got != c["mask"]).float().mean()) <= tol, (c["shape"], c["upsample"], dtype)
    mc = torch.randn(3, 32, 50)
    idx = torch.tensor([7, 0, 49, 13], dtype=torch.int64)
    got = postprocess.gather_mask_coefficients(mc.to(dev), 1, idx.to(dev)).cpu()
    assert torch.equal(got, mc[1][:, idx].t())


def test_process_mask_on_the_emulator(post, golden_dir):
    mask_kernel_checks(golden_dir)


def test_segmentation_predict_flow(post, emu, golden_dir):
    """Model -> NMS -> coefficient gather -> process_mask -> scale_boxes, glued the way SegmentationPredictor does it
    (models/yolo/segment/predict.py), against the oracle's pieces applied to the REAL reference's head outputs."""
    import json

    from oracle import nms_ref, post_ref
    from tests.helpers import fill_by_name
    from yolo_master_amd import postprocess
    from yolo_master_amd.nms import non_max_suppression
    from yolo_master_amd.nn.tasks import DetectionModel

    z = np.load(golden_dir / "fwd_seg_n.npz")
    cfg = json.loads(str(z["cfg"]))
    m = DetectionModel(cfg)
    full = dict(m.state_dict())
    full.update(fill_by_name(json.loads(str(z["spec"])), seed=11, gain=0.8))
    m.load_state_dict(full)
    m.eval()
    x = torch.from_numpy(z["x"])
    H, W = x.shape[2:]
    with torch.inference_mode():
        y, preds = m._predict_once(x)
    conf, iou = 0.3, 0.6
    dets, idx = non_max_suppression(y, conf, iou, return_idxs=True)
    ref_y, ref_p = torch.from_numpy(z["y"]), torch.from_numpy(z["proto"])
    nc = m.model[-1].nc
    rd, ri = nms_ref.non_max_suppression(ref_y[:, : 4 + nc].numpy(), conf, iou, return_idxs=True)
    total = 0
    for b in range(x.shape[0]):
        assert np.array_equal(idx[b].numpy(), ri[b]), "kept anchors differ from the oracle NMS on the reference's predictions"
        n = len(ri[b])
        total += n
        coefs = postprocess.gather_mask_coefficients(preds["mask_coefficient"], b, idx[b].contiguous())
        masks = postprocess.process_mask(preds["proto"][b], coefs, dets[b][:, :4].contiguous(), (H, W), upsample=True)
        ref_masks = post_ref.process_mask(ref_p[b], ref_y[b, 4 + nc:, ri[b]].t().contiguous(), torch.from_numpy(rd[b][:, :4].copy()), (H, W), upsample=True)
        assert masks.shape == ref_masks.shape == (n, H, W)
        if n:
            assert float((masks != ref_masks).float().mean()) <= 2e-3
        boxes = postprocess.scale_boxes((H, W), dets[b][:, :4].clone(), (2 * H + 3, 2 * W))
        assert np.allclose(boxes.numpy(), post_ref.scale_boxes((H, W), rd[b][:, :4], (2 * H + 3, 2 * W)), atol=2e-3)
    assert total > 0, "the fixture should keep some detections at this threshold"


def match_kernel_checks(dev="cpu", golden_dir=None):
    """ymk_box_iou / ymk_match_predictions (all fixture cases as ONE batch, padded like nms_padded's output) against the real
    reference's vectors; shared with tests/test_gpu_next.py."""
    from tests.test_oracle_post import match_cases
    from yolo_master_amd import postprocess

    cs = [c for c in match_cases(golden_dir) if not c["tied"]]
    for c in cs:   # box_iou, bit-exact
        if c["labels"].shape[0] and c["dets"].shape[0]:
            got = postprocess.box_iou(torch.from_numpy(c["labels"][:, 1:].copy()).to(dev), torch.from_numpy(c["dets"][:, :4].copy()).to(dev))
            assert np.array_equal(got.cpu().numpy(), c["iou"])
    B, max_det, T = len(cs), 300, len(cs[0]["iouv"])
    dets = torch.zeros((B, max_det, 6))
    counts = torch.zeros((B,), dtype=torch.int32)
    labels, off = [], [0]
    for b, c in enumerate(cs):
        n = c["dets"].shape[0]
        dets[b, :n] = torch.from_numpy(c["dets"])
        dets[b, n:, :4] = torch.tensor([10.0, 10.0, 200.0, 200.0])      # padding rows must not take part
        counts[b] = n
        labels.append(torch.from_numpy(c["labels"]))
        off.append(off[-1] + c["labels"].shape[0])
    labels = torch.cat(labels).contiguous()
    got = postprocess.match_predictions(dets.to(dev), counts.to(dev), labels.to(dev), torch.tensor(off, dtype=torch.int32).to(dev),
                                        torch.from_numpy(cs[0]["iouv"]).to(dev)).cpu().numpy()
    assert got.shape == (B, max_det, T)
    for b, c in enumerate(cs):
        n = c["dets"].shape[0]
        assert np.array_equal(got[b, :n], c["correct"]), f"image {b}: correct matrix differs from the reference"
        assert not got[b, n:].any()
    # per-image validation records (update_metrics): batched product path vs the restatement of the reference's per-image loop
    from oracle import post_ref

    stats = postprocess.batch_stats(dets.to(dev), counts.to(dev), labels.to(dev), torch.tensor(off, dtype=torch.int32).to(dev),
                                    torch.from_numpy(cs[0]["iouv"]).to(dev))
    want = post_ref.batch_stats([c["dets"] for c in cs], [c["labels"] for c in cs], cs[0]["iouv"])
    assert len(stats) == len(want) == B
    for b, (g, w) in enumerate(zip(stats, want)):
        for key in ("tp", "conf", "pred_cls", "target_cls", "target_img"):
            assert g[key].shape == w[key].shape and np.array_equal(g[key], w[key]), f"image {b}: {key}"
        assert np.array_equal(g["tp"], cs[b]["correct"].reshape(g["tp"].shape))


def test_validation_matching_kernels_on_emulator(post, golden_dir):
    match_kernel_checks("cpu", golden_dir)
