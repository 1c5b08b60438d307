"""How far does the REFERENCE's own reduced-precision evaluation sit from its fp32 one?  BASELINE config 3 (YOLO-Master-S, 64 x 3 x
640 x 640, the benchmarked configuration) run by the real reference model on CPU in fp32 (= tests/golden/fwd_s640_b64.npz), in fp16
(`model.half()`: the reference's `half=True` mode, engine/predictor.py:174,415) and in bf16, same weights and images:

    python tests/golden/make_golden_ref16.py            (build container only: needs /root/reference; ~10 min of CPU)

For each 16-bit format the script records, against the reference's fp32 result: routed-expert agreement per (image, layer), score and
box error percentiles on the images whose routing agrees, NMS kept-set Jaccard per image, and — for every decision that flipped — the
fp32 router's margins (logit gap between the 2nd and 3rd expert, distance of the 2nd expert's weight from the 0.4 retention
threshold, moe/modules.py:676-678).  tests/test_gpu_baseline_configs.py holds libymk's 16-bit results to THESE numbers: a 16-bit
implementation is as good as it can be when it is no further from the fp32 reference than the reference's own 16-bit evaluation.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import refboot  # noqa: E402
from yolo_master_amd.weights import CFG_DIR, synth_input, synth_state_dict  # noqa: E402

refboot.boot()
from ultralytics.nn.tasks import DetectionModel as RefModel  # noqa: E402
from ultralytics.utils.nms import non_max_suppression as ref_nms  # noqa: E402

REF_YAML = "/root/reference/ultralytics/cfg/models/master/v0/det/yolo-master-s.yaml"
MOE = (3, 6, 9, 12)
THRESH = 0.4


def retained_from_route_weights(rw: torch.Tensor) -> torch.Tensor:
    """ES_MOE._sparse_forward (moe/modules.py:665-684): the spatially constant routing weights ARE the importances; the top-ranked
    expert is always retained, a lower-ranked selected expert when its importance reaches the threshold."""
    rw = rw.float()
    return (rw > 0) & ((rw == rw.max(1, keepdim=True).values) | (rw >= THRESH))


def run(ref, x, dtype):
    routes = {}
    hooks = [ref.model[i].routing.register_forward_hook(lambda mod, inp, o, idx=i: routes.__setitem__(idx, o[:, :, 0, 0].float().clone())) for i in MOE]
    with torch.inference_mode():
        y = ref(x.to(dtype))
        y = (y[0] if isinstance(y, (tuple, list)) else y).float()
    for h in hooks:
        h.remove()
    return y, routes


if __name__ == "__main__":
    torch.set_num_threads(8)
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    z = np.load(HERE / "fwd_s640_b64.npz")
    B, H, W, seed, conf, iou = int(z["B"]), int(z["H"]), int(z["W"]), int(z["seed"]), float(z["conf"]), float(z["iou"])
    ref = RefModel(REF_YAML, ch=3, nc=80, verbose=False)
    ref.load_state_dict(synth_state_dict(ref.state_dict(), seed=0, calib=str(CFG_DIR / str(z["calib"]))))
    ref.eval()
    ref.fuse(verbose=False)
    x = synth_input(B, H, W, seed=seed)[:nimg]
    B = nimg
    y32, r32 = run(ref, x, torch.float32)
    idx = z["y_idx"].astype(np.int64)
    A, ch = y32.shape[2], y32.shape[1]
    keep = idx < B * ch * A
    idx = idx[keep]
    d32 = np.abs(y32.reshape(-1)[torch.from_numpy(idx)].numpy() - z["y_val"][keep]).max()
    assert d32 == 0 or (nimg < 64 and d32 < 1e-2), f"this run must reproduce the committed fp32 fixture (max |d| {d32})"   # a sub-batch changes oneDNN's blocking
    for i in MOE:
        assert np.array_equal(retained_from_route_weights(r32[i]).numpy(), z[f"route{i}_retained"][:B]), f"retained-set derivation, layer {i}"
    with torch.inference_mode():
        k32 = [ref_nms(y32[b:b + 1].clone(), conf, iou, return_idxs=True, max_time_img=1e9)[1][0].reshape(-1).numpy() for b in range(B)]
    rec = {"B": B, "conf": conf, "iou": iou}
    import copy
    for tag, dtype in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        m16 = copy.deepcopy(ref).to(dtype)
        import time
        t0 = time.time()
        y16, r16 = run(m16, x, dtype)
        print(f"[{tag}] reference forward in {dtype}: {time.time() - t0:.1f} s, finite {bool(torch.isfinite(y16).all())}")
        same_img = np.ones(B, bool)
        agree = 0
        flips = []
        for i in MOE:
            ret16, ret32 = retained_from_route_weights(r16[i]).numpy(), z[f"route{i}_retained"][:B]
            same = (ret16 == ret32).all(1)
            same_img &= same
            agree += int(same.sum())
            lg = z[f"route{i}_logits"][:B]
            srt = np.sort(lg, 1)[:, ::-1]
            gw = z[f"route{i}_route_w"][:B]
            second = np.sort(gw, 1)[:, -2]
            for b in np.nonzero(~same)[0]:
                flips.append((b, i, float(srt[b, 1] - srt[b, 2]), float(abs(second[b] - THRESH))))
        g = y16.reshape(-1)[torch.from_numpy(idx)].numpy()
        c, img = (idx // A) % ch, idx // (A * ch)
        err = np.abs(g - z["y_val"][keep])
        sel = same_img[img]
        ps = np.percentile(err[(c >= 4) & sel], [50, 99, 100])
        pb = np.percentile(err[(c < 4) & sel], [50, 99, 100])
        with torch.inference_mode():
            k16 = [ref_nms(y16[b:b + 1].clone(), conf, iou, return_idxs=True, max_time_img=1e9)[1][0].reshape(-1).numpy() for b in range(B)]
        jac = np.array([len(set(a.tolist()) & set(b_.tolist())) / max(len(set(a.tolist()) | set(b_.tolist())), 1) if (len(a) or len(b_)) else 1.0
                        for a, b_ in zip(k16, k32)])
        print(f"[{tag}] reference-{tag} vs reference-fp32: routing identical on {agree}/{4 * B} (image, layer) pairs = {int(same_img.sum())}/{B} images; "
              f"same-routing images: scores p50 {ps[0]:.2e} p99 {ps[1]:.2e} max {ps[2]:.2e}, boxes px p50 {pb[0]:.2e} p99 {pb[1]:.2e} max {pb[2]:.2e}; "
              f"kept-set Jaccard median {np.median(jac[same_img]):.3f} min {jac[same_img].min():.3f} (all images: median {np.median(jac):.3f} min {jac.min():.3f})")
        for b, i, gap, thr in flips:
            print(f"[{tag}]   flipped: image {b} layer {i}: fp32 logit gap 2nd-3rd {gap:.3e}, |w2 - 0.4| {thr:.3e}")
        rec.update({f"{tag}::agree_pairs": agree, f"{tag}::same_img": same_img, f"{tag}::score_pct": ps, f"{tag}::box_pct": pb, f"{tag}::jaccard": jac,
                    f"{tag}::flips": np.array(flips, np.float64).reshape(-1, 4)})
    # margins of ALL fp32 decisions (the histogram the flips are read against)
    for i in MOE:
        lg = np.sort(z[f"route{i}_logits"][:B], 1)[:, ::-1]
        rec[f"margin::gap{i}"] = (lg[:, 1] - lg[:, 2]).astype(np.float32)
        rec[f"margin::thr{i}"] = np.abs(np.sort(z[f"route{i}_route_w"][:B], 1)[:, -2] - THRESH).astype(np.float32)
    if nimg == 64:
        np.savez_compressed(HERE / "ref16_s640_b64.npz", **rec)
        print("wrote", (HERE / "ref16_s640_b64.npz").stat().st_size, "bytes")
