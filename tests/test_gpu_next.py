"""Opt-in tiled convolution core (csrc/conv_glds.hip, include/ymk_next.h) on the GPU: the cases of
tests/test_hostemu_conv.py through the real library, plus the library's own dispatch with YMK_ENABLE set.
First hardware run: round 2 (profiles/r02_first_hw_run.log)."""
import os
import subprocess
import sys

import pytest
import torch

from tests.test_hostemu_conv import BIG_TILE_CASES, CASES, CAT2_CASES, EPILOGUE_CASES, EXPERT_CASES, run_case, run_cat2_case, run_expert_case, tile_flags

pytestmark = pytest.mark.gpu

BIG = [(4, 80, 80, 128, 128, 3, 2, True, False, False, 0, 0, 0), (4, 40, 40, 256, 256, 3, 2, True, True, False, 0, 0, 1),
       (2, 80, 80, 512, 128, 1, 1, True, False, False, 0, 128, 0), (8, 40, 40, 256, 64, 3, 1, True, False, True, 0, 0, 0)]


@pytest.mark.parametrize("case", CASES + BIG)
def test_conv2d_glds_direct(case):
    from yolo_master_amd import _lib

    run_case(_lib.load(), case, dev="cuda:0", stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [c[:-1] + (tile_flags(*t),) for c, t in BIG_TILE_CASES] + [
    (8, 80, 80, 256, 256, 3, 2, True, False, False, 0, 0, tile_flags(256, 208)),      # the detector's 256 -> 256 stride-2 row: 62 tiles, ragged last
    (16, 40, 40, 384, 256, 1, 1, True, False, False, 0, 0, tile_flags(256, 208)),     # a C3k2 tail at 40^2
    (16, 20, 20, 768, 512, 1, 1, True, True, False, 0, 0, tile_flags(256, 208))])     # two cout tiles per pixel tile + residual
def test_conv2d_glds_forced_big_tiles(case):
    """The one-workgroup-per-CU tiles (128 x 512, 256 x 256, 128 x 256 and round 6's 256 x 208) forced through the entry point's flags."""
    from yolo_master_amd import _lib

    run_case(_lib.load(), case, dev="cuda:0", stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", EPILOGUE_CASES + [(4, 80, 80, 256, 512, 1, 1, "gelu", False, False, 0, 0, None),      # the MoT token FFN at config 5's P4
                                                   (8, 40, 40, 512, 1024, 1, 1, "gelu", False, False, 0, 0, None)])
def test_conv2d_gelu_and_sigmoid_epilogues(case):
    from yolo_master_amd import _lib

    run_case(_lib.load(), case, dev="cuda:0", stream=torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("case", CAT2_CASES + [(4, 80, 80, 256, 256, 128, True, True, 0, 0, 0, 0), (4, 40, 40, 512, 256, 256, True, True, 0, 0, 0, 1)])
def test_conv1x1_cat2_glds_direct(case):
    from yolo_master_amd import _lib

    run_cat2_case(_lib.load(), case, dev="cuda:0", stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", EXPERT_CASES + [(4, 80, 80, 128, 256, 3, 8, [[0, 7], [3, 3], [5, 1], [2, 6]], 0)])
def test_expert_conv_glds_direct(case):
    from yolo_master_amd import _lib

    run_expert_case(_lib.load(), case, dev="cuda:0", stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


@pytest.mark.parametrize("enable", ["1", "3"])
def test_model_with_the_new_core_enabled(enable):
    """The S detector at batch 8 with YMK_ENABLE set (read once per process -> subprocess): same detections as the default
    kernels on the median image (bf16; the two paths differ only in fp32 summation order)."""
    code = (
        "import torch, json\n"
        "from yolo_master_amd.nn.tasks import DetectionModel\n"
        "from yolo_master_amd.weights import synth_state_dict, synth_input\n"
        "m = DetectionModel('yolo-master-s.yaml'); m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))\n"
        "m = m.eval().to('cuda:0').set_compute_dtype(torch.bfloat16)\n"
        "with torch.inference_mode(): y, _ = m._predict_once(synth_input(8, 640, 640, seed=3).to('cuda:0'))\n"
        "torch.save(y.cpu(), __import__('sys').argv[1])\n")
    outs = []
    for tag, env in (("base", {}), ("glds", {"YMK_ENABLE": enable})):
        path = f"/tmp/ymk_next_{tag}_{enable}.pt"
        subprocess.run([sys.executable, "-c", code, path], check=True, cwd=str(__import__('pathlib').Path(__file__).resolve().parent.parent), env={**os.environ, **env}, timeout=600)
        outs.append(torch.load(path))
    d = (outs[0] - outs[1]).abs()
    assert float(d[:, 4:].median()) < 2e-3 and float(d[:, :4].median()) < 0.5, (float(d[:, 4:].median()), float(d[:, :4].median()))


def test_scale_boxes_vs_reference_golden(golden_dir):
    """Bit-exact against the real reference's vectors (tests/golden/make_golden_post.py), single image and batched."""
    import numpy as np

    from tests.test_oracle_post import cases
    from yolo_master_amd import postprocess

    cs = list(cases(golden_dir))
    for c in cs:
        boxes = torch.from_numpy(c["boxes"].copy()).to("cuda:0")
        postprocess.scale_boxes(c["img1"], boxes, c["img0"], ratio_pad=c["ratio_pad"], padding=c["padding"], xywh=c["xywh"])
        assert np.array_equal(boxes[:, :4].cpu().numpy(), c["out"]), (c["img1"], c["img0"])
    sel = [c for c in cs if c["img1"] == (640, 640) and c["padding"] and not c["xywh"] and c["ratio_pad"] is None]
    dets = torch.full((len(sel), 50, 6), -3.0)
    for b, c in enumerate(sel):
        dets[b, :37] = torch.from_numpy(c["boxes"])
    counts = torch.tensor([37, 20, 0, 37, 5][: len(sel)], dtype=torch.int32)
    d = dets.to("cuda:0")
    postprocess.scale_detections((640, 640), d, counts.to("cuda:0"), [c["img0"] for c in sel])
    for b, c in enumerate(sel):
        n = int(counts[b])
        assert np.array_equal(d[b, :n, :4].cpu().numpy(), c["out"][:n]) and torch.equal(d[b, n:].cpu(), dets[b, n:])


def test_segment_kernels_and_head(golden_dir):
    """Segment head on the GPU: the two layout kernels against their contracts, then the v0 seg-n model against the REAL
    reference's SegmentationModel output (tests/golden/make_golden_seg.py)."""
    import json

    import numpy as np

    from tests.helpers import fill_by_name
    from tests.test_hostemu_post import seg_kernel_checks
    from yolo_master_amd import ops
    from yolo_master_amd.nn.tasks import DetectionModel

    seg_kernel_checks("cuda:0")
    z = np.load(golden_dir / "fwd_seg_n.npz")
    cfg = json.loads(str(z["cfg"]))
    m = DetectionModel(cfg)
    full = dict(m.state_dict())
    full.update(fill_by_name(json.loads(str(z["spec"])), seed=11, gain=0.8))
    m.load_state_dict(full)
    m.eval().to("cuda:0")
    with torch.inference_mode():
        y, preds = m._predict_once(torch.from_numpy(z["x"]).to("cuda:0"))
        ycat = torch.cat([y, preds["mask_coefficient"]], 1).cpu()
        proto = ops.nhwc_to_nchw_f32(preds["proto"]).cpu()
    ref_y, ref_p = torch.from_numpy(z["y"]), torch.from_numpy(z["proto"])
    assert float((ycat[:, :4] - ref_y[:, :4]).abs().max()) <= 1e-3 + 1e-4 * float(ref_y[:, :4].abs().max())
    assert float((ycat[:, 4:] - ref_y[:, 4:]).abs().max()) <= 1e-4 * max(1.0, float(ref_y[:, 4:].abs().max()))
    assert float((proto - ref_p).abs().max()) <= 1e-4 * max(1.0, float(ref_p.abs().max()))


def test_process_mask_vs_reference_golden(golden_dir):
    from tests.test_hostemu_post import mask_kernel_checks

    mask_kernel_checks(golden_dir, "cuda:0")


@pytest.mark.parametrize("case", __import__("tests.test_hostemu_pre", fromlist=["CASES"]).CASES)
def test_letterbox_preprocess_vs_oracle(case):
    """Device-side LetterBox + BGR->RGB + CHW + /255 (csrc/preproc.hip) bit-exact against oracle/pre_ref.py."""
    from tests.test_hostemu_pre import run_case as run_pre_case
    from yolo_master_amd import _lib

    run_pre_case(_lib.load(), case, dev="cuda:0", stream=torch.cuda.current_stream().cuda_stream)


def test_preprocess_wrapper_feeds_the_model():
    """yolo_master_amd.preprocess.preprocess on host uint8 images -> the detector's input tensor; same detections as feeding the
    oracle's pre-processed tensor."""
    import numpy as np

    from oracle import pre_ref
    from yolo_master_amd.preprocess import preprocess

    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (90, 120, 3), dtype=np.uint8), rng.integers(0, 256, (128, 64, 3), dtype=np.uint8)]
    x = preprocess(imgs, (128, 128), device="cuda:0")
    assert x.shape == (2, 3, 128, 128) and x.dtype == torch.float32
    assert np.array_equal(x.cpu().numpy(), pre_ref.preprocess(imgs, (128, 128)))


def test_validation_matching_vs_reference_golden(golden_dir):
    """box_iou + batched match_predictions on the GPU against the real reference's vectors (tests/golden/make_golden_match.py)."""
    from tests.test_hostemu_post import match_kernel_checks

    match_kernel_checks("cuda:0", golden_dir)
