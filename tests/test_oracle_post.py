"""scale_boxes / clip_boxes oracle against golden vectors from the REAL reference
(tests/golden/make_golden_post.py).  CPU only."""
import numpy as np

from oracle import post_ref


def cases(golden_dir):
    z = np.load(golden_dir / "post_scale.npz")
    for i in range(int(z["n"])):
        m = z[f"c{i}_meta"]
        rp = z[f"c{i}_rp"]
        yield dict(img1=(int(m[0]), int(m[1])), img0=(int(m[2]), int(m[3])), padding=bool(m[4]), xywh=bool(m[5]),
                   ratio_pad=((float(rp[0]), float(rp[0])), (float(rp[1]), float(rp[2]))) if m[6] else None,
                   boxes=z[f"c{i}_boxes"], out=z[f"c{i}_out"])


def test_scale_boxes_oracle_is_bit_exact(golden_dir):
    n = 0
    for c in cases(golden_dir):
        got = post_ref.scale_boxes(c["img1"], c["boxes"][:, :4], c["img0"], ratio_pad=c["ratio_pad"], padding=c["padding"], xywh=c["xywh"])
        assert np.array_equal(got, c["out"]), (c["img1"], c["img0"])
        if not c["xywh"]:
            assert (got[:, [0, 2]] >= 0).all() and (got[:, [0, 2]] <= c["img0"][1]).all() and (got[:, [1, 3]] <= c["img0"][0]).all()
        n += 1
    assert n == 10


def test_letterbox_params_round_half_even():
    # 500x375 into 640x384: gain 1.024, width fills, rows padded 0 / 1 -> the -0.1 nudge decides
    assert post_ref.letterbox_params((384, 640), (375, 500)) == (1.024, 64, 0)
    assert post_ref.letterbox_params((640, 640), (480, 640)) == (1.0, 0, 80)


def mask_cases(golden_dir):
    import torch

    z = np.load(golden_dir / "post_mask.npz")
    for i in range(int(z["n"])):
        m = z[f"c{i}_meta"]
        ms = tuple(int(v) for v in z[f"c{i}_mshape"])
        mask = np.unpackbits(z[f"c{i}_mask"])[: int(np.prod(ms))].reshape(ms) if np.prod(ms) else np.zeros(ms, np.uint8)
        yield dict(protos=torch.from_numpy(z[f"c{i}_protos"]), coefs=torch.from_numpy(z[f"c{i}_coefs"]), boxes=torch.from_numpy(z[f"c{i}_boxes"]),
                   shape=(int(m[0]), int(m[1])), upsample=bool(m[2]), mask=torch.from_numpy(mask))


def test_process_mask_oracle_matches_reference(golden_dir):
    n = 0
    for c in mask_cases(golden_dir):
        got = post_ref.process_mask(c["protos"], c["coefs"], c["boxes"], c["shape"], upsample=c["upsample"])
        assert got.shape == c["mask"].shape
        # bit-identical at generation time; a different thread count may move a sum by an ulp and flip a pixel that sits at 0
        assert float((got != c["mask"]).float().mean()) <= 1e-4 if got.numel() else True
        n += 1
    assert n == 6


def match_cases(golden_dir):
    z = np.load(golden_dir / "post_match.npz")
    for i in range(int(z["n"])):
        yield {"dets": z[f"c{i}_dets"], "labels": z[f"c{i}_labels"], "iou": z[f"c{i}_iou"], "correct": z[f"c{i}_correct"],
               "tied": bool(z[f"c{i}_tied"]), "iouv": z["iouv"]}


def test_validation_matching_restatement_matches_reference(golden_dir):
    """oracle box_iou / match_predictions against vectors from the real reference (metrics.box_iou, BaseValidator.match_predictions)."""
    n = 0
    for c in match_cases(golden_dir):
        iou = post_ref.box_iou(c["labels"][:, 1:], c["dets"][:, :4])
        assert np.array_equal(iou, c["iou"])
        got = post_ref.match_predictions(c["dets"][:, 5], c["labels"][:, 0], iou, c["iouv"])
        assert np.array_equal(got, c["correct"])
        n += 1
    assert n >= 7
