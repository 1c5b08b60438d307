"""Oracle vs the real reference at BASELINE configs[4]'s own configuration (v0_10 MoA + MoT detector, L scale, 1280 x 1280): the
restatement (oracle/model_ref + gated_ref / moa_ref / mot_ref) regenerates the fixture of tests/golden/make_golden_cfg5_l.py —
dense samples of y, every routing decision, NMS kept indices and the Cluster-Weighted boxes.  The generator itself asserted
bit-equality of every layer against the reference; this test re-checks the committed vectors without the reference checkout
(~40 s of CPU: the one large-size oracle run of the CPU suite)."""
import numpy as np
import torch

from oracle import model_ref, nms_ref


def test_oracle_regenerates_the_l_scale_1280_fixture(golden_dir):
    from tests.test_gpu_mixture import load_cfg5_l

    z, cfg, rcp, sd, x = load_cfg5_l(golden_dir, "base")
    torch.set_num_threads(rcp["threads"])
    info = {}
    with torch.inference_mode():
        y, _, _ = model_ref.forward(cfg, sd, x, fused=False, moe_info=info)
    got = y.reshape(-1)[torch.from_numpy(z["base::y_idx"])].numpy()
    np.testing.assert_allclose(got, z["base::y_val"], rtol=0, atol=1e-5)   # bit-equal at the generator's thread count; 1e-5 allows another host's oneDNN blocking
    for key in [f for f in z.files if f.startswith("base::route::")]:
        ind = info[key[len("base::route::"):]]["indices"].numpy()
        assert np.array_equal(ind.astype(np.int8), z[key]), key
    dets, idx = nms_ref.non_max_suppression(y.numpy(), rcp["conf"], rcp["iou"], return_idxs=True)
    for b in range(y.shape[0]):
        assert np.array_equal(idx[b], z[f"base::nms_idx{b}"])
        np.testing.assert_allclose(dets[b], z[f"base::nms_det{b}"], rtol=0, atol=1e-4)
        yb = y[b].numpy()
        conf, cls = yb[4:].max(0), yb[4:].argmax(0)
        m = conf > np.float32(rcp["conf"])
        cands = np.concatenate([nms_ref.xywh2xyxy(yb[:4].T.copy())[m], conf[m, None], cls[m, None].astype(np.float32)], 1).astype(np.float32)
        pos = {int(a): j for j, a in enumerate(np.arange(yb.shape[1])[m])}
        keep = np.array([pos[int(a)] for a in idx[b]], np.int64)
        cw = nms_ref.cw_refine(cands, keep, rcp["iou"], rcp["sigma"])
        np.testing.assert_allclose(cw, z[f"base::cw_box{b}"], rtol=0, atol=1e-3)
