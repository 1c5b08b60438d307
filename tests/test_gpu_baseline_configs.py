"""Parity at BASELINE.json's own sizes, against vectors produced by the REAL reference (tests/golden/make_golden.py baseline):

* config 2 — YOLO-Master-N, 32 x 3 x 640 x 640, fp32: routing decisions, NMS kept anchor indices and classes bit-exact;
  scores within 1e-4; boxes within 1e-4 in the unit the network regresses (DFL bins = pixels / anchor stride);
  every layer's activations within 1e-4.  No noise-relative escape, no overlap-ratio fallback.
* config 3 — YOLO-Master-S, 64 x 3 x 640 x 640, bf16 (the benchmarked configuration) and fp16 (the reference's half=True
  precision) against the reference's fp32 result, with the REFERENCE'S OWN 16-bit evaluation as the bar
  (tests/golden/make_golden_ref16.py): routing agreement, score / box error percentiles, kept-set overlap.

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
    eb_rel = float((err[ch < 4] / np.maximum(np.abs(z["y_val"][ch < 4]), 1.0)).max())    # relative to the coordinate (pixels up to 640)
    print(f"config 2: worst layer |d| {worst[0]:.2e} (layer {worst[1]}); y scores {es:.2e}, boxes {eb_px:.2e} px = {eb_bins:.2e} bins = {eb_rel:.2e} relative "
          f"(reference fp32 vs fp64: {float(z['y_noise_box']):.2e} px, {float(z['y_noise_cls']):.2e})")
    assert es <= 1e-4, f"scores {es:.3e}"
    # north_star: "box coords ... within 1e-4 fp32".  Stated three ways, all asserted: in DFL bins (the decode's own unit: error / stride,
    # independent of the pyramid level) <= 1e-4; relative to the coordinate (max(|value|, 1 px)) <= 1e-4; and in pixels <= 1e-3 with the
    # reference's OWN fp32 evaluation-order noise floor printed beside it (its fp32-vs-fp64 distance on this fixture, 6e-4 px) — the
    # absolute pixel figure (4-5e-4 px) is above 1e-4 because a stride-32 bin is 32 px wide, not because of lost digits
    assert eb_bins <= 1e-4, f"boxes {eb_bins:.3e} bins ({eb_px:.3e} px)"
    assert eb_rel <= 1e-4, f"boxes {eb_rel:.3e} relative"
    assert eb_px <= 1e-3, f"boxes {eb_px:.3e} px (the reference's own noise floor: {float(z['y_noise_box']):.3e} px)"
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


@pytest.mark.parametrize("fmt", ["bf16", "f16"])
def test_config3_s_b64_640_16bit_vs_reference(fmt, golden_dir):
    """The benchmarked configuration, in both 16-bit formats (bf16 = the bench line; f16 = the reference's own `half=True`
    precision on libymk_f16.so).  A 16-bit evaluation cannot meet an fp32 bar through 26 layers; the bar that CAN be stated is the
    reference's own: tests/golden/make_golden_ref16.py ran the REAL reference model in fp16 and in bf16 on the same weights and
    images and recorded how far each sits from the reference's fp32 result (routing agreement, score / box error percentiles on
    same-routing images, NMS kept-set Jaccard, and the fp32 router margins of every flipped decision).  libymk's 16-bit result must
    be AT LEAST AS CLOSE to the reference's fp32 result as the reference's own evaluation in that format is (x 1.25 for the
    percentile estimates: the two evaluations round at different points of the graph)."""
    from yolo_master_amd.nms import non_max_suppression
    from yolo_master_amd.weights import synth_input

    dtype = torch.bfloat16 if fmt == "bf16" else torch.float16
    z = load_npz(golden_dir / "fwd_s640_b64.npz")
    r16 = load_npz(golden_dir / "ref16_s640_b64.npz")
    B, H, W = int(z["B"]), int(z["H"]), int(z["W"])
    assert (B, H, W, chr(int(z["scale"]))) == (64, 640, 640, "s") and int(r16["B"]) == B
    m = _model("s", dtype, str(z["calib"]))
    x = synth_input(B, H, W, seed=int(z["seed"]))
    with torch.inference_mode():
        y, _ = m._predict_once(x.to(DEV))
    m.check_flags()
    assert torch.isfinite(y).all()
    same_img = np.ones(B, bool)
    agree, total, rw_err = 0, 0, 0.0
    flipped = []
    for i in MOE:
        r = m.model[i].last_route
        same = ((r["gate_w"] > 0).cpu().numpy() == z[f"route{i}_retained"]).all(1)
        same_img &= same
        agree, total = agree + int(same.sum()), total + B
        rw_err = max(rw_err, float(np.abs(r["route_w"].cpu().numpy() - z[f"route{i}_route_w"]).max()))
        flipped += [(int(b), i, float(r16[f"margin::gap{i}"][b]), float(r16[f"margin::thr{i}"][b])) for b in np.nonzero(~same)[0]]
    yc = y.cpu()
    A = yc.shape[2]
    idx = z["y_idx"].astype(np.int64)
    g = yc.reshape(-1)[torch.from_numpy(idx)].numpy()
    ch, img = (idx // A) % yc.shape[1], idx // (A * yc.shape[1])
    err = np.abs(g - z["y_val"])
    sel = same_img[img]                       # samples that lie in images whose four routed expert sets equal the reference's
    ps = np.percentile(err[(ch >= 4) & sel], [50, 99, 100])
    pb = np.percentile(err[(ch < 4) & sel], [50, 99, 100])
    dets, kept = non_max_suppression(y, float(z["conf"]), float(z["iou"]), return_idxs=True)
    jac = []
    for b in range(B):
        a_, b_ = set(kept[b].cpu().numpy().tolist()), set(z[f"nms{b}_idx"].tolist())
        jac.append(len(a_ & b_) / max(len(a_ | b_), 1) if (a_ or b_) else 1.0)
    jac = np.array(jac)
    ref_agree, ref_ps, ref_pb, ref_jac = int(r16[f"{fmt}::agree_pairs"]), r16[f"{fmt}::score_pct"], r16[f"{fmt}::box_pct"], r16[f"{fmt}::jaccard"]
    ref_same = r16[f"{fmt}::same_img"]
    print(f"config 3 ({fmt} vs reference fp32) libymk | the reference's own {fmt}: routing identical on {agree} | {ref_agree} of {total} (image, layer) pairs "
          f"= {int(same_img.sum())} | {int(ref_same.sum())} of {B} images (route_w max err {rw_err:.2e}); same-routing images: scores p50 {ps[0]:.2e} | "
          f"{ref_ps[0]:.2e}, p99 {ps[1]:.2e} | {ref_ps[1]:.2e}; boxes px p50 {pb[0]:.2e} | {ref_pb[0]:.2e}, p99 {pb[1]:.2e} | {ref_pb[1]:.2e}; kept-set "
          f"Jaccard median {np.median(jac[same_img]):.3f} | {np.median(ref_jac[ref_same]):.3f}, min {jac[same_img].min():.3f} | {ref_jac[ref_same].min():.3f}")
    for b, i, gap, thr in flipped:
        print(f"   flipped: image {b} layer {i}: fp32 router margins: logit gap 2nd-3rd {gap:.2e}, |w2 - 0.4| {thr:.2e}")
    # Every image's FIRST flipped layer (later ones see a different input) is a near call of the fp32 router itself: its margin is
    # inside the range in which the reference's own evaluation in this format flips (first-flip margins there: fp16 <= 0.022, bf16 <= 0.051)
    first = {}
    for b, i, gap, thr in flipped:
        if b not in first or i < first[b][0]:
            first[b] = (i, min(gap, thr))
    ref_first = {}
    for b, i, gap, thr in r16[f"{fmt}::flips"]:
        if b not in ref_first or i < ref_first[b][0]:
            ref_first[b] = (i, min(gap, thr))
    lim = 1.5 * max(v[1] for v in ref_first.values())
    assert all(v[1] <= lim for v in first.values()), f"a routing decision flipped far from the fp32 router's thresholds: {first} (limit {lim:.3f})"
    assert agree >= ref_agree - 4, f"{fmt}: routing agrees on {agree} pairs; the reference's own {fmt} run on {ref_agree}"
    assert ps[0] <= 1.25 * ref_ps[0] and ps[1] <= 1.25 * ref_ps[1], f"scores {ps} vs the reference's own {ref_ps}"
    assert pb[0] <= 1.25 * ref_pb[0] and pb[1] <= 1.25 * ref_pb[1], f"boxes {pb} vs the reference's own {ref_pb}"
    assert np.median(jac[same_img]) >= np.median(ref_jac[ref_same]) - 0.02 and jac[same_img].min() >= ref_jac[ref_same].min() - 0.1


def _forward_nms(m, x):
    from yolo_master_amd.nms import nms_padded

    with torch.inference_mode():
        y, _ = m._predict_once(x)
        dets, counts, idx, _ = nms_padded(y, 0.25, 0.7, max_det=300)
    m.check_flags()
    return y, dets, counts, idx


def test_config1_full_batch_images_are_independent():
    """Size-independent property at the benchmarked size (S, 64 x 3 x 640 x 640, bf16; no fixture can hold this batch's outputs): the path
    partitions over images — the routed experts group images by expert, tiles of the convolution cores straddle image boundaries, NMS runs a
    workgroup per image — so every image's result must be BIT-identical whatever shares the batch with it: (1) the batch in another order
    gives the same per-image outputs and detections, (2) eight of the images alone (other tile shapes, other expert groups) give the same."""
    from yolo_master_amd.weights import synth_input

    m = _model("s", torch.bfloat16, "cond_s.npz")
    x = synth_input(64, 640, 640, seed=3).to(DEV)
    y, dets, counts, idx = _forward_nms(m, x)
    assert int(counts.sum()) > 0
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(7)).to(DEV)
    y2, dets2, counts2, idx2 = _forward_nms(m, x[perm].contiguous())
    assert torch.equal(y2, y[perm]), "per-image head outputs depend on the image's position in the batch"
    assert torch.equal(counts2, counts[perm]) and torch.equal(idx2, idx[perm]) and torch.equal(dets2, dets[perm])
    sub = perm[:8]
    y3, dets3, counts3, idx3 = _forward_nms(m, x[sub].contiguous())
    assert torch.equal(y3, y[sub]), "eight images alone differ from the same images inside the batch of 64"
    assert torch.equal(counts3, counts[sub]) and torch.equal(idx3, idx[sub]) and torch.equal(dets3, dets[sub])


def test_config5_full_size_images_are_independent():
    """The same property at BASELINE config 5's own size (MoA + MoT YAML at the L scale, 1280 x 1280, fp16 — the reference's `half`): four
    images together and two of them alone, per-token routing, chunked GroupNorm / channel statistics (chunk counts depend on the map, not on
    the batch), whole-map attention over 6400 tokens, Cluster-Weighted NMS."""
    from yolo_master_amd import ops
    from yolo_master_amd.nms import nms_padded
    from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load
    from yolo_master_amd.weights import synth_input, synth_state_dict

    dtype = torch.float16 if ops.HAS_F16 else torch.bfloat16
    cfg = yaml_model_load("yolo-master-moa-mot.yaml")
    cfg.setdefault("scales", {}).setdefault("l", yaml_model_load("yolo-master.yaml")["scales"]["l"])
    cfg["scale"] = "l"
    m = DetectionModel(cfg)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
    m = m.eval().to(DEV).set_compute_dtype(dtype)
    x = synth_input(4, 1280, 1280, seed=5).to(DEV)

    def run(xx):
        with torch.inference_mode():
            y, _ = m._predict_once(xx)
            out = nms_padded(y, 0.05, 0.7, max_det=300, cluster=True, sigma=0.1)
        m.check_flags()
        return (y, *out[:3])

    y, dets, counts, idx = run(x)
    assert bool(torch.isfinite(y.float()).all())
    sel = torch.tensor([2, 0], device=DEV)
    y2, dets2, counts2, idx2 = run(x[sel].contiguous())
    assert torch.equal(y2, y[sel]), "two images alone differ from the same images inside the batch of four"
    assert torch.equal(counts2, counts[sel]) and torch.equal(idx2, idx[sel]) and torch.equal(dets2, dets[sel])
