"""Golden vectors for oracle/post_ref.py from the REAL reference (`ultralytics.utils.ops.scale_boxes`).
Run in the build container (needs /root/reference):  python tests/golden/make_golden_post.py
Writes tests/golden/post_scale.npz and asserts that the oracle is bit-exact on every case."""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import post_ref, refboot  # noqa: E402

refboot.boot()
from ultralytics.utils.ops import scale_boxes  # noqa: E402

if __name__ == "__main__":
    g = torch.Generator().manual_seed(99)
    rec, n = {}, 0
    cases = [((640, 640), (480, 640), None, True, False), ((640, 640), (1080, 1920), None, True, False), ((384, 640), (375, 500), None, True, False),
             ((640, 640), (333, 1000), None, True, False), ((1280, 1280), (2160, 3840, 3), None, True, False), ((640, 640), (640, 640), None, True, False),
             ((640, 480), (1333, 1000), None, False, False), ((640, 640), (427, 640), None, True, True),
             ((640, 640), (480, 640), ((0.75, 0.75), (12.0, 80.0)), True, False), ((96, 160), (1, 7), None, True, False)]
    for img1, img0, rp, padding, xywh in cases:
        boxes = torch.rand(37, 4, generator=g) * torch.tensor([img1[1], img1[0], img1[1], img1[0]]) * 1.2 - 20.0   # some out of frame
        boxes = torch.cat([boxes, torch.rand(37, 2, generator=g)], 1)
        ref = scale_boxes(img1, boxes[:, :4].clone(), img0, ratio_pad=rp, padding=padding, xywh=xywh).numpy()
        got = post_ref.scale_boxes(img1, boxes[:, :4].numpy(), img0, ratio_pad=rp, padding=padding, xywh=xywh)
        assert np.array_equal(ref, got), (img1, img0, np.abs(ref - got).max())
        rec[f"c{n}_boxes"], rec[f"c{n}_out"] = boxes.numpy(), ref
        rec[f"c{n}_meta"] = np.array([img1[0], img1[1], img0[0], img0[1], int(padding), int(xywh), 0 if rp is None else 1], np.int32)
        rec[f"c{n}_rp"] = np.array([0, 0, 0] if rp is None else [rp[0][0], rp[1][0], rp[1][1]], np.float64)
        n += 1
    rec["n"] = np.int32(n)
    np.savez_compressed(HERE / "post_scale.npz", **rec)
    print(f"post_scale.npz: {n} cases, oracle bit-exact vs the reference on all")
