"""Golden vectors for oracle/post_ref.process_mask from the REAL reference (`ultralytics.utils.ops.process_mask`).
Run in the build container:  python tests/golden/make_golden_mask.py   -> tests/golden/post_mask.npz
(masks stored bit-packed; the oracle is asserted bit-identical on every case)."""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import post_ref, refboot  # noqa: E402

refboot.boot()
from ultralytics.utils.ops import process_mask  # noqa: E402

if __name__ == "__main__":
    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(77)
    rec, n = {}, 0
    for (mh, mw), shape, nd, up in (((40, 40), (160, 160), 7, True), ((40, 40), (160, 160), 7, False), ((24, 16), (96, 64), 5, True),
                                    ((20, 28), (80, 112), 9, False), ((40, 40), (160, 160), 0, True), ((12, 12), (48, 48), 3, True)):
        protos = torch.randn(32, mh, mw, generator=g)
        coefs = torch.randn(nd, 32, generator=g) * 0.5
        cx, cy = torch.rand(nd, generator=g) * shape[1], torch.rand(nd, generator=g) * shape[0]
        w, h = torch.rand(nd, generator=g) * shape[1] * 0.6 + 2, torch.rand(nd, generator=g) * shape[0] * 0.6 + 2
        boxes = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
        ref = process_mask(protos.clone(), coefs.clone(), boxes.clone(), shape, upsample=up)
        got = post_ref.process_mask(protos, coefs, boxes, shape, upsample=up)
        assert torch.equal(ref, got), (mh, mw, shape, up)
        rec[f"c{n}_protos"], rec[f"c{n}_coefs"], rec[f"c{n}_boxes"] = protos.numpy(), coefs.numpy(), boxes.numpy()
        rec[f"c{n}_meta"] = np.array([shape[0], shape[1], int(up), nd], np.int32)
        rec[f"c{n}_mask"] = np.packbits(ref.numpy().reshape(-1))
        rec[f"c{n}_mshape"] = np.array(ref.shape, np.int32)
        n += 1
    rec["n"] = np.int32(n)
    np.savez_compressed(HERE / "post_mask.npz", **rec)
    print(f"post_mask.npz: {n} cases, oracle bit-identical to the reference on all")
