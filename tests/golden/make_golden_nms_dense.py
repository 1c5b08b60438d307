"""Golden vectors for the validator's NMS on dense scenes, from the REAL reference (`ultralytics.utils.nms.non_max_suppression`,
pure-torch TorchNMS path): conf 0.001, multi_label=True (utils/nms.py:119-123) on prediction tensors in which every (anchor, class)
pair is a candidate — 8400 x 80 = 672 000 per image, far above max_nms = 30 000, so the reference's sort-and-truncate
(utils/nms.py:142-146) decides which 30 000 reach the greedy pass.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_nms_dense.py

y is regenerated from its seed (tests/helpers.dense_pred); the fixture stores the kept anchor indices and detections.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import nms_ref, refboot  # noqa: E402
from tests.helpers import dense_pred  # noqa: E402

refboot.boot()
from ultralytics.utils.nms import non_max_suppression as ref_nms  # noqa: E402

CASES = {   # name: (B, nc, A, seed, kwargs)
    "val640": (2, 80, 8400, 640, dict(conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300)),
    "single1280": (1, 80, 33600, 1280, dict(conf_thres=0.001, iou_thres=0.7, max_det=300)),   # best-class candidates: 33 600 > max_nms
}

if __name__ == "__main__":
    torch.set_num_threads(8)
    rec = {}
    for name, (B, nc, A, seed, kw) in CASES.items():
        y = dense_pred(B, nc, A, seed, frame=float(seed))
        outs, idxs = [], []
        with torch.inference_mode():
            for b in range(B):   # one call per image: the reference's wall-clock limit (utils/nms.py:167-169) would drop later images
                o, k = ref_nms(y[b:b + 1].clone(), return_idxs=True, max_time_img=1e9, **kw)
                outs.append(o[0].numpy()); idxs.append(k[0].reshape(-1).numpy())
        mine, mine_i = nms_ref.non_max_suppression(y.numpy(), return_idxs=True, **kw)
        for b in range(B):
            ncand = int((y[b, 4:] > kw["conf_thres"]).sum()) if kw.get("multi_label") else int((y[b, 4:].amax(0) > kw["conf_thres"]).sum())
            ok = np.array_equal(mine_i[b], idxs[b]) and np.array_equal(mine[b], outs[b])
            print(f"[nms_dense {name}] image {b}: {ncand} candidates, kept {len(idxs[b])}; numpy oracle == reference: {ok}")
            assert ok and ncand > 30000
            rec[f"{name}::idx{b}"], rec[f"{name}::dets{b}"] = idxs[b].astype(np.int64), outs[b]
        rec[f"{name}::recipe"] = np.array([B, nc, A, seed])
    np.savez_compressed(HERE / "nms_dense.npz", **rec)
    print("wrote", (HERE / "nms_dense.npz").stat().st_size, "bytes")
