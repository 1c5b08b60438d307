"""Oracle (numpy restatement of utils/nms.py) vs the real reference on the validator's dense-scene settings: conf 0.001,
multi_label, 672 000 candidates per image against max_nms = 30 000 (tests/golden/make_golden_nms_dense.py)."""
import numpy as np

from oracle import nms_ref


def test_oracle_truncates_to_max_nms_like_the_reference(golden_dir):
    from tests.helpers import dense_pred
    from tests.test_gpu_kernels import DENSE_KW

    z = np.load(golden_dir / "nms_dense.npz")
    for case, kw in DENSE_KW.items():
        B, nc, A, seed = [int(v) for v in z[f"{case}::recipe"]]
        y = dense_pred(B, nc, A, seed, frame=float(seed))
        dets, idx = nms_ref.non_max_suppression(y.numpy(), return_idxs=True, **kw)
        for b in range(B):
            assert np.array_equal(idx[b], z[f"{case}::idx{b}"]) and np.array_equal(dets[b], z[f"{case}::dets{b}"]), (case, b)
