"""The oracle restatement (oracle/model_ref.py, oracle/nms_ref.py) against golden vectors produced by the
REAL reference (tests/golden/make_golden.py).  Runs on CPU; this is what pins the oracle."""
import numpy as np
import pytest
import torch

from tests.helpers import load_npz

CASES = ["n640", "n_ragged", "n_tiny", "s_small", "l_tiny", "n640_b32"]   # s640_b64 (config 3): same code path, 40 s on CPU


@pytest.mark.parametrize("case", CASES)
def test_forward_restatement_matches_reference(case, golden_dir):
    from oracle import model_ref, nms_ref
    from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load
    from yolo_master_amd.weights import CFG_DIR, synth_input, synth_state_dict

    z = load_npz(golden_dir / f"fwd_{case}.npz")
    scale = chr(int(z["scale"]))
    B, H, W, seed = int(z["B"]), int(z["H"]), int(z["W"]), int(z["seed"])
    cfg = yaml_model_load(f"yolo-master-{scale}.yaml")
    sd = synth_state_dict(DetectionModel(cfg).state_dict(), seed=0, calib=str(CFG_DIR / str(z["calib"])) if "calib" in z else "auto")
    taps, info = {}, {}
    with torch.inference_mode():
        y, _, _ = model_ref.forward(cfg, sd, synth_input(B, H, W, seed=seed), taps=taps, moe_info=info)
    # bit-exact on the machine that generated the fixtures; other CPUs may pick other oneDNN kernels,
    # so the gate is a tight tolerance (routing decisions and NMS indices stay exact).
    for i in range(25):
        got = taps[i].reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        np.testing.assert_allclose(got, z[f"layer{i}_val"], rtol=1e-4, atol=1e-5, err_msg=f"layer {i}")
    got = y.reshape(-1)[torch.from_numpy(z["y_idx"].astype(np.int64))].numpy()
    np.testing.assert_allclose(got, z["y_val"], rtol=1e-4, atol=1e-5)
    same_bits = np.array_equal(got, z["y_val"])   # false when this run's CPU kernels summed in another order (thread count)
    for i in (3, 6, 9, 12):
        assert np.array_equal(info[f"model.{i}"]["retained"].numpy(), z[f"route{i}_retained"])
        np.testing.assert_allclose(info[f"model.{i}"]["route_w"].numpy(), z[f"route{i}_route_w"], atol=1e-6)
    dets, idx = nms_ref.non_max_suppression(y.numpy(), float(z["conf"]), float(z["iou"]), return_idxs=True)
    for b in range(B):
        if bool(z["ties"][b]):
            continue
        if same_bits:
            assert np.array_equal(idx[b], z[f"nms{b}_idx"])
            np.testing.assert_allclose(dets[b], z[f"nms{b}_dets"], rtol=1e-4, atol=1e-4)
        else:   # ulp-level differences in y may flip a near-threshold IoU / score decision: compare the kept sets
            a, r = set(idx[b].tolist()), set(z[f"nms{b}_idx"].tolist())
            assert len(a & r) >= 0.97 * len(a | r), f"image {b}: kept sets differ beyond near-threshold flips"   # (both empty: equal)


@pytest.mark.parametrize("case", ["single", "multi", "agnostic", "caps", "empty", "one", "classes", "classes_multi",
                                  "seg", "seg_multi", "seg_caps"])      # seg*: nc + carried mask rows (utils/nms.py:76-81,117)
def test_nms_restatement_matches_reference(case, golden_dir):
    from oracle import nms_ref

    z = load_npz(golden_dir / f"nms_{case}.npz")
    kw = dict(conf_thres=float(z["arg_conf_thres"]), iou_thres=float(z["arg_iou_thres"]),
              multi_label=bool(z["arg_multi_label"]), agnostic=bool(z["arg_agnostic"]), max_det=int(z["arg_max_det"]),
              max_nms=int(z["arg_max_nms"]), classes=z["arg_classes"].tolist() if "arg_classes" in z else None,
              nc=int(z["arg_nc"]) if "arg_nc" in z else 0)
    dets, idx = nms_ref.non_max_suppression(z["y"], return_idxs=True, **kw)
    for b in range(z["y"].shape[0]):
        assert np.array_equal(idx[b], z[f"idx{b}"]), f"{case} image {b}"
        assert np.array_equal(dets[b], z[f"dets{b}"]), f"{case} image {b}"


def test_cw_refine_properties():
    """Invariants of the CW-NMS spec (the implementation itself is pinned against the reference C++ in test_oracle_cw.py)."""
    from oracle import nms_ref

    rng = np.random.default_rng(0)
    n = 200
    xy = rng.uniform(50, 300, (n, 2)).astype(np.float32)
    wh = rng.uniform(30, 90, (n, 2)).astype(np.float32)
    cands = np.concatenate([xy - wh / 2, xy + wh / 2, rng.uniform(0.3, 0.9, (n, 1)).astype(np.float32),
                            rng.integers(0, 3, (n, 1)).astype(np.float32)], 1)
    keep = nms_ref.nms_greedy(cands[:, :4] + cands[:, 5:6] * 7680, cands[:, 4], 0.5)
    ref = nms_ref.cw_refine(cands, keep, 0.5, 0.1)
    assert ref.shape == (len(keep), 4)
    # a survivor whose cluster is only itself keeps its box exactly (weight s*exp(0) on itself)
    lonely = nms_ref.cw_refine(cands[:1], np.array([0]), 0.5, 0.1)
    np.testing.assert_allclose(lonely[0], cands[0, :4].astype(np.float64), rtol=1e-12)
    # refined boxes stay inside the hull of their cluster
    assert (ref[:, 0] >= cands[:, 0].min() - 1e-6).all() and (ref[:, 2] <= cands[:, 2].max() + 1e-6).all()


@pytest.mark.parametrize("case", ["sparse", "dense", "disabled", "all", "k3of4"])
def test_esmoe_modes_restatement_matches_reference(case, golden_dir):
    """oracle.model_ref.es_moe in every dispatch mode + the eval-time state, against the real reference module's vectors."""
    from oracle import model_ref

    z = load_npz(golden_dir / f"esmoe_{case}.npz")
    kw = {"sparse": {}, "dense": dict(sparse=False), "disabled": dict(sparse=False), "all": dict(top_k=4, hard_top_k=False, sparse=False),
          "k3of4": dict(top_k=3, thr=0.2)}[case]
    sd = {f"m.{k}": torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}
    info = {}
    with torch.inference_mode():
        y = model_ref.es_moe(sd, "m", torch.from_numpy(z["x"]), info=info, **kw)
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-5, atol=1e-6)
    r = info["m"]
    assert np.array_equal(r["retained"].numpy(), z["retained"])
    np.testing.assert_allclose(r["usage"].numpy(), z["usage"], atol=1e-7)
    np.testing.assert_allclose(float(r["lb_loss"]), float(z["lb_loss"]), atol=1e-6)
