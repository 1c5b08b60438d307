"""Parity at BASELINE.json's own sizes, against vectors produced by the REAL reference (tests/golden/make_golden.py baseline):

* config 2 — YOLO-Master-N, 32 x 3 x 640 x 640, fp32: routing decisions, NMS kept anchor indices and classes bit-exact;
  scores within 1e-4; boxes within 1e-4 in the unit the network regresses (DFL bins = pixels / anchor stride);
  every layer's activations within 1e-4.  No noise-relative escape, no overlap-ratio fallback.
* config 3 — YOLO-Master-S, 64 x 3 x 640 x 640, bf16 (the benchmarked configuration) against the reference's fp32
  result: routing agreement, score / box error percentiles, kept-set overlap.

Both use the well-conditioned synthetic weights of tools/make_conditioned.py (cfg/cond_<scale>.npz): with them the
reference's own fp32 result is 6e-4 px / 2.4e-6 from the exact (fp64) one at 640 x 640, and its discrete decisions
are stable under an evaluation-order perturbation (recorded per image in the fixture as `stable`)."""
import numpy as np
import pytest
import torch

from tests.helpers import load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MOE = (3, 6, 9, 12)


def _model(scale, dtype, calib):
    from yolo_master_amd.nn.tasks import DetectionModel
    from yolo_master_amd.weights import CFG_DIR, synth_state_dict

    m = DetectionModel(f"yolo-master-{scale}.yaml")
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0, calib=str(CFG_DIR / calib)))
    return m.eval().to(DEV).set_compute_dtype(dtype)


def _anchor_stride(a, H, W):
    """Stride of anchor index a (levels 8, 16, 32 in that order, head.py:186-194 / tal.py:398-411)."""
    n8, n16 = (H // 8) * (W // 8), (H // 16) * (W // 16)
    return np.where(a < n8, 8.0, np.where(a < n8 + n16, 16.0, 32.0))


def test_config2_n_b32_640_fp32_vs_reference(golden_dir):
    from yolo_master_amd import ops
    from yolo_master_amd.nms import non_max_suppression
    from yolo_master_amd.weights import synth_input

    z = load_npz(golden_dir / "fwd_n640_b32.npz")
    B, H, W = int(z["B"]), int(z["H"]), int(z["W"])
    assert (B, H, W, chr(int(z["scale"]))) == (32, 640, 640, "n")
    m = _model("n", torch.float32, str(z["calib"]))
    x = synth_input(B, H, W, seed=int(z["seed"]))
    taps = {}
    with torch.inference_mode():
        y, _ = m._predict_once(x.to(DEV), taps=taps)
    m.check_flags()
    assert bool(z["route_stable"].all()) and bool(z["stable"].all()), "fixture must be decision-stable on all 32 images"
    for i in MOE:   # routed expert sets: identical to the reference on every image; weights 1e-4
        r = m.model[i].last_route
        assert np.array_equal((r["gate_w"] > 0).cpu().numpy(), z[f"route{i}_retained"]), f"layer {i}: retained experts differ"
        assert np.abs(r["route_w"].cpu().numpy() - z[f"route{i}_route_w"]).max() <= 1e-4
        assert np.abs(r["gate_w"].cpu().numpy() - z[f"route{i}_gate_w"]).max() <= 1e-4
        # eval-time state of the module after this batch (modules.py:706-741), values from the reference's own buffers
        assert np.abs(m.model[i].expert_usage_counts.cpu().numpy() - z[f"route{i}_usage"]).max() <= 1e-5, f"layer {i}: expert_usage_counts"
        assert abs(float(m.model[i].load_balancing_loss) - float(z[f"route{i}_lbloss"])) <= 1e-5, f"layer {i}: load_balancing_loss"
    worst = (0, 0.0)
    for i in range(25):   # every layer against the reference's fp32 values, flat 1e-4
        t = taps[i]
        if not torch.is_tensor(t):
            t = t.materialise()
        got = ops.nhwc_to_nchw_f32(t).cpu().reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        e = float(np.abs(got - z[f"layer{i}_val"]).max())
        worst = max(worst, (e, i))
        assert e <= 1e-4, f"layer {i}: |hip - reference| = {e:.3e}"
    yc = y.cpu()
    A = yc.shape[2]
    idx = z["y_idx"].astype(np.int64)
    g = yc.reshape(-1)[torch.from_numpy(idx)].numpy()
    ch, a = (idx // A) % yc.shape[1], idx % A
    err = np.abs(g - z["y_val"])
    es = float(err[ch >= 4].max())
    eb_px = float(err[ch < 4].max())
    eb_bins = float((err[ch < 4] / _anchor_stride(a[ch < 4], H, W)).max())
    print(f"config 2: worst layer |d| {worst[0]:.2e} (layer {worst[1]}); y scores {es:.2e}, boxes {eb_px:.2e} px = {eb_bins:.2e} bins "
          f"(reference fp32 vs fp64: {float(z['y_noise_box']):.2e} px, {float(z['y_noise_cls']):.2e})")
    assert es <= 1e-4, f"scores {es:.3e}"
    assert eb_bins <= 1e-4, f"boxes {eb_bins:.3e} bins ({eb_px:.3e} px)"
    dets, kept = non_max_suppression(y, float(z["conf"]), float(z["iou"]), return_idxs=True)
    nk = 0
    for b in range(B):
        ref_idx, ref_d = z[f"nms{b}_idx"], z[f"nms{b}_dets"]
        got_idx, d = kept[b].cpu().numpy(), dets[b].cpu().numpy()
        assert np.array_equal(got_idx, ref_idx), f"image {b}: kept anchor indices differ from the reference"
        assert np.array_equal(d[:, 5], ref_d[:, 5]), f"image {b}: classes differ"
        assert np.abs(d[:, 4] - ref_d[:, 4]).max(initial=0) <= 1e-4, f"image {b}: scores"
        s = _anchor_stride(ref_idx, H, W)[:, None]
        assert (np.abs(d[:, :4] - ref_d[:, :4]) <= 1e-4 * s).all(), f"image {b}: boxes {np.abs(d[:, :4] - ref_d[:, :4]).max():.3e} px"
        nk += len(ref_idx)
    print(f"config 2: {nk} kept detections over {B} images identical to the reference (indices, classes)")


def test_config3_s_b64_640_bf16_vs_reference_fp32(golden_dir):
    """The benchmarked configuration.  bf16 storage (8 mantissa bits) through 26 layers cannot meet an fp32 bar; what is
    asserted is how far the bf16 result sits from the REFERENCE's fp32 one, image by image (bounds = 2x measured on MI355X)."""
    from yolo_master_amd.nms import non_max_suppression
    from yolo_master_amd.weights import synth_input

    z = load_npz(golden_dir / "fwd_s640_b64.npz")
    B, H, W = int(z["B"]), int(z["H"]), int(z["W"])
    assert (B, H, W, chr(int(z["scale"]))) == (64, 640, 640, "s")
    m = _model("s", torch.bfloat16, str(z["calib"]))
    x = synth_input(B, H, W, seed=int(z["seed"]))
    with torch.inference_mode():
        y, _ = m._predict_once(x.to(DEV))
    m.check_flags()
    assert torch.isfinite(y).all()
    same_img = np.ones(B, bool)
    agree, total, rw_err = 0, 0, 0.0
    for i in MOE:
        r = m.model[i].last_route
        same = ((r["gate_w"] > 0).cpu().numpy() == z[f"route{i}_retained"]).all(1)
        same_img &= same
        agree, total = agree + int(same.sum()), total + B
        rw_err = max(rw_err, float(np.abs(r["route_w"].cpu().numpy() - z[f"route{i}_route_w"]).max()))
    yc = y.cpu()
    A = yc.shape[2]
    idx = z["y_idx"].astype(np.int64)
    g = yc.reshape(-1)[torch.from_numpy(idx)].numpy()
    ch, img = (idx // A) % yc.shape[1], idx // (A * yc.shape[1])
    err = np.abs(g - z["y_val"])
    sel = same_img[img]                       # samples that lie in images whose four routed expert sets equal the reference's
    ps = np.percentile(err[(ch >= 4) & sel], [50, 99, 100])
    pb = np.percentile(err[(ch < 4) & sel], [50, 99, 100])
    ps_all = np.percentile(err[ch >= 4], [50, 99, 100])
    pb_all = np.percentile(err[ch < 4], [50, 99, 100])
    dets, kept = non_max_suppression(y, float(z["conf"]), float(z["iou"]), return_idxs=True)
    jac = []
    for b in range(B):
        a_, b_ = set(kept[b].cpu().numpy().tolist()), set(z[f"nms{b}_idx"].tolist())
        jac.append(len(a_ & b_) / max(len(a_ | b_), 1) if (a_ or b_) else 1.0)
    jac = np.array(jac)
    print(f"config 3 (bf16 vs reference fp32): routing identical on {agree}/{total} (image, layer) pairs = {int(same_img.sum())}/{B} images, "
          f"route_w max err {rw_err:.2e}; same-routing images: scores |d| p50 {ps[0]:.2e} p99 {ps[1]:.2e} max {ps[2]:.2e}, boxes px p50 "
          f"{pb[0]:.2e} p99 {pb[1]:.2e} max {pb[2]:.2e}, kept-set Jaccard median {np.median(jac[same_img]):.3f} min {jac[same_img].min():.3f}; "
          f"all images: scores p50 {ps_all[0]:.2e} p99 {ps_all[1]:.2e}, boxes p50 {pb_all[0]:.2e} p99 {pb_all[1]:.2e}, Jaccard median "
          f"{np.median(jac):.3f} min {jac.min():.3f}")
    assert agree >= 0.8 * total, f"bf16 changed the routed expert set on {total - agree} of {total} (image, layer) pairs"
    # measured on MI355X (round 2): 225/256 pairs = 49/64 images; same-routing images: scores p50 7.5e-5 p99 3.8e-3, boxes p50 0.14 px
    # p99 1.3 px, Jaccard median 0.865 (min 0.78); bounds = 2x those
    assert ps[0] <= 1.5e-4 and ps[1] <= 8e-3, f"scores {ps}"
    assert pb[0] <= 0.3 and pb[1] <= 2.7, f"boxes {pb}"
    assert np.median(jac[same_img]) >= 0.75 and jac[same_img].min() >= 0.5, f"kept-set Jaccard {np.median(jac[same_img])} min {jac[same_img].min()}"
