"""Pre-processing oracle (oracle/pre_ref.py): LetterBox geometry against the REAL reference's get_params (tests/golden/pre_params.json),
the restated OpenCV 8-bit bilinear against its invariants, and the product's host-side geometry against the same vectors."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pre_ref


def _opts(o):
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in o.items()}


def test_letterbox_params_match_the_reference(golden_dir):
    from yolo_master_amd.preprocess import letterbox_params as product_params

    rows = json.load(open(golden_dir / "pre_params.json"))
    assert len(rows) >= 150
    for r in rows:
        kw = {"new_shape": (640, 640), **_opts(r["opt"])}
        for fn in (pre_ref.letterbox_params, product_params):
            p = fn(tuple(r["shape"]), **kw)
            assert list(p["new_unpad"]) == r["new_unpad"] and [p["top"], p["bottom"], p["left"], p["right"]] == \
                [r["top"], r["bottom"], r["left"], r["right"]], (fn.__module__, r)
            assert list(p["ratio"]) == r["ratio"]


def test_resize_invariants():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(pre_ref.resize_linear_u8(img, (53, 37)), img)                       # identity
    for dsize in ((80, 60), (20, 11), (53, 74), (106, 37), (7, 5)):
        const = np.full((37, 53, 3), 201, np.uint8)
        assert (pre_ref.resize_linear_u8(const, dsize) == 201).all()                          # constants are preserved
        out = pre_ref.resize_linear_u8(img, dsize)
        assert out.shape == (dsize[1], dsize[0], 3) and out.dtype == np.uint8
        assert out.min() >= img.min() and out.max() <= img.max()                              # convex combination of taps
    ramp = np.tile(np.arange(0, 200, 4, dtype=np.uint8)[None, :, None], (9, 1, 3))            # monotone stays monotone
    up = pre_ref.resize_linear_u8(ramp, (131, 9)).astype(int)
    assert (np.diff(up[4, :, 0]) >= 0).all()
    big = rng.integers(0, 256, (64, 48, 3), dtype=np.uint8)                                   # exact 2x: rounded 2x2 mean
    want = (big.astype(int).reshape(32, 2, 24, 2, 3).sum((1, 3)) + 2) >> 2
    assert np.array_equal(pre_ref.resize_linear_u8(big, (24, 32)), want.astype(np.uint8))


@pytest.mark.parametrize("dsize", [(640, 480), (427, 640), (100, 333), (900, 601)])
def test_resize_agrees_with_float_bilinear_within_one_level(dsize):
    """Same sampling geometry as torch's bilinear (align_corners=False, no antialias); fixed point differs by at most one level."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (375, 500, 3), dtype=np.uint8)
    out = pre_ref.resize_linear_u8(img, dsize).astype(np.float32)
    ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), size=(dsize[1], dsize[0]), mode="bilinear",
                        align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(out - ref).max() <= 1.0 + 1e-3


def test_preprocess_layout_and_scale():
    rng = np.random.default_rng(2)
    imgs = [rng.integers(0, 256, (48, 64, 3), dtype=np.uint8), rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)]
    x = pre_ref.preprocess(imgs, (64, 64))
    assert x.shape == (2, 3, 64, 64) and x.dtype == np.float32
    assert x[0, 0, 0, 0] == np.float32(114) / np.float32(255)                                  # padding rows on top of the 48-row image
    assert np.array_equal(x[1, 0], imgs[1][..., 2].astype(np.float32) / np.float32(255))      # channel 0 of the output is R = BGR[2]
    assert np.array_equal(x[0, 2, 8:56], imgs[0][..., 0].astype(np.float32) / np.float32(255))
