"""Golden vectors for validation matching from the REAL reference: `ultralytics.utils.metrics.box_iou` and
`BaseValidator.match_predictions` (engine/validator.py:301-336) on seeded detections / labels -> tests/golden/post_match.npz.
Run in the build container:  python tests/golden/make_golden_match.py"""
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle import post_ref, refboot  # noqa: E402

refboot.boot()
from ultralytics.engine.validator import BaseValidator  # noqa: E402
from ultralytics.utils.metrics import box_iou  # noqa: E402

if __name__ == "__main__":
    g = torch.Generator().manual_seed(7)
    iouv = torch.linspace(0.5, 0.95, 10)
    self_ = types.SimpleNamespace(iouv=iouv)
    rec, n = {"iouv": iouv.numpy()}, 0
    for D, L, nc, jitter in ((60, 12, 3, 6.0), (300, 40, 5, 10.0), (17, 1, 1, 3.0), (5, 30, 2, 4.0), (0, 4, 2, 1.0), (25, 0, 2, 1.0), (120, 25, 1, 15.0)):
        gt_xy = torch.rand(L, 2, generator=g) * 500 + 20
        gt_wh = torch.rand(L, 2, generator=g) * 120 + 20
        gt = torch.cat([gt_xy - gt_wh / 2, gt_xy + gt_wh / 2], 1)
        gcls = torch.randint(0, nc, (L,), generator=g).float()
        if L and D:   # detections = jittered copies of random labels (several per label) + some clutter, confidence-sorted
            src = torch.randint(0, L, (D,), generator=g)
            boxes = gt[src] + (torch.rand(D, 4, generator=g) - 0.5) * jitter * torch.rand(D, 1, generator=g) * 4
            pcls = torch.where(torch.rand(D, generator=g) < 0.85, gcls[src], torch.randint(0, nc, (D,), generator=g).float())
        else:
            boxes = torch.rand(D, 4, generator=g) * 300
            boxes[:, 2:] += boxes[:, :2]
            pcls = torch.randint(0, nc, (D,), generator=g).float()
        conf = torch.sort(torch.rand(D, generator=g), descending=True)[0]
        iou = box_iou(gt, boxes)
        o_iou = post_ref.box_iou(gt.numpy(), boxes.numpy())
        assert np.array_equal(iou.numpy(), o_iou), f"box_iou oracle differs on case {n}"
        correct = BaseValidator.match_predictions(self_, pcls, gcls, iou).numpy()
        o_corr = post_ref.match_predictions(pcls.numpy(), gcls.numpy(), o_iou, iouv.numpy())
        assert np.array_equal(correct, o_corr), f"match_predictions oracle differs on case {n}"
        # tie check: two labels with the same IoU for one detection make the reference's result depend on numpy's unstable sort
        m = (iou * (gcls[:, None] == pcls)).numpy()
        tied = any(len(np.unique(col[col >= 0.5])) != (col >= 0.5).sum() for col in m.T) if L and D else False
        rec[f"c{n}_dets"] = torch.cat([boxes, conf[:, None], pcls[:, None]], 1).numpy()
        rec[f"c{n}_labels"] = torch.cat([gcls[:, None], gt], 1).numpy()
        rec[f"c{n}_iou"], rec[f"c{n}_correct"], rec[f"c{n}_tied"] = iou.numpy(), correct, np.bool_(tied)
        print(f"case {n}: D={D} L={L}: true positives at 0.5 / 0.75 / 0.95 = {correct[:, 0].sum()} / {correct[:, 5].sum()} / {correct[:, 9].sum()}; ties={tied}")
        n += 1
    rec["n"] = np.int32(n)
    np.savez_compressed(HERE / "post_match.npz", **rec)
    print(f"post_match.npz: {n} cases; oracle bit-exact vs the reference on all")
