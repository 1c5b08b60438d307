"""BASELINE.json configs[4] at ITS OWN PRECISION: "fp16 + CW-NMS".  How far does the REFERENCE's own fp16 evaluation (`model.half()`:
its `half=True` mode, engine/predictor.py:174,415) of the L-scale MoA + MoT detector at 1280 x 1280 sit from its fp32 one?  Same
network, weights and images as tests/golden/fwd_cfg5_l.npz (rebuilt from that fixture), real reference on CPU:

    python tests/golden/make_golden_cfg5_l_ref16.py        (build container only: needs /root/reference)

Records, against the reference's fp32 result (the method of make_golden_ref16.py, config 3): score / box error percentiles of y, the
gated blocks' routed experts per image, the share of per-token top-k selections that agree per MoT block, the NMS kept-set Jaccard per
image (conf 0.25, IoU 0.7) and the distance of the Cluster-Weighted boxes (sigma 0.1) of the common survivors.
tests/test_gpu_mixture.py::test_config5_fp16_vs_the_references_own_half_run holds libymk_f16 to THESE numbers.
"""
import copy
import json
import sys
import time
import warnings
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import nms_ref, refboot  # noqa: E402
from tests.helpers import condition_bn, fill_by_name  # noqa: E402
from yolo_master_amd.weights import synth_input  # noqa: E402

refboot.boot()
from ultralytics.nn.tasks import DetectionModel as RefModel  # noqa: E402

PCT = (50, 90, 99)


def cw_boxes(y, conf, iou, sigma):
    """Greedy NMS + Cluster-Weighted boxes of every image by the numpy oracle (pinned to the reference: test_oracle_golden / test_oracle_cw)."""
    dets, idx = nms_ref.non_max_suppression(y, conf, iou, return_idxs=True)
    out = []
    for b in range(y.shape[0]):
        yb = y[b]
        c, cl = yb[4:].max(0), yb[4:].argmax(0)
        mk = c > np.float32(conf)
        cands = np.concatenate([nms_ref.xywh2xyxy(yb[:4].T.copy())[mk], c[mk, None], cl[mk, None].astype(np.float32)], 1).astype(np.float32)
        pos = {int(a): j for j, a in enumerate(np.arange(y.shape[2])[mk])}
        keep = np.array([pos[int(a)] for a in idx[b]], np.int64)
        out.append((dets[b], idx[b], nms_ref.cw_refine(cands, keep, iou, sigma) if len(keep) else np.zeros((0, 4))))
    return out


def run(ref, x, dtype):
    routes = {}

    def grab(name):
        def hook(mod, inp, out):
            # gated routers return (weights, indices, stats); MoT routers (weights [B,E,H,W], ...)
            o = out if torch.is_tensor(out) else out[0]
            ind = out[1] if (isinstance(out, (tuple, list)) and len(out) > 1 and torch.is_tensor(out[1]) and not out[1].is_floating_point()) else None
            routes[name] = (o.detach().float().clone(), None if ind is None else ind.detach().clone())
        return hook

    hs = []
    for name, mod in ref.named_modules():
        cls = type(mod).__name__
        if cls in ("_MoTRouter", "EfficientSpatialRouter", "DualStreamRouter", "LowRankDualStreamRouter") or (name.endswith(".routing") and "MoE" in type(ref.get_submodule(name.rsplit(".", 1)[0])).__name__):
            hs.append(mod.register_forward_hook(grab(name)))
    t0 = time.time()
    with torch.inference_mode():
        y = ref(x.to(dtype))
        y = (y[0] if isinstance(y, (tuple, list)) else y).float()
    for h in hs:
        h.remove()
    print(f"[cfg5_l_ref16] reference forward in {dtype}: {time.time() - t0:.1f} s, {len(routes)} routers hooked")
    return y, routes


if __name__ == "__main__":
    torch.set_num_threads(8)
    z = np.load(HERE / "fwd_cfg5_l.npz")
    cfg, rcp = json.loads(str(z["cfg"])), json.loads(str(z["recipe"]))
    sd = fill_by_name(json.loads(str(z["spec"])), seed=5, gain=1.0)
    condition_bn(sd)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    x = synth_input(rcp["batch"], rcp["img"], rcp["img"], seed=rcp["x_seed"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = RefModel(copy.deepcopy(cfg), ch=3, nc=80, verbose=False).eval()
    ref.load_state_dict(sd)
    y32, r32 = run(ref, x, torch.float32)
    yi = torch.from_numpy(z["base::y_idx"])
    assert torch.equal(y32.reshape(-1)[yi], torch.from_numpy(z["base::y_val"])), "this is not the network of fwd_cfg5_l.npz"
    y16, r16 = run(copy.deepcopy(ref).half(), x, torch.float16)
    A = y32.shape[2]
    img = rcp["img"]
    stride_of = torch.cat([torch.full(((img // s) ** 2,), float(s)) for s in (8, 16, 32)])
    d = (y16 - y32).abs()
    rec = {"recipe": np.array(json.dumps({**rcp, "format": "fp16", "percentiles": PCT}))}
    rec["score_err_pct"] = np.percentile(d[:, 4:].numpy(), PCT)
    rec["box_err_px_pct"] = np.percentile(d[:, :4].numpy(), PCT)
    rec["box_err_bins_pct"] = np.percentile((d[:, :4] / stride_of).numpy(), PCT)
    print("[cfg5_l_ref16] |y16 - y32| percentiles", PCT, ": scores", rec["score_err_pct"], " boxes px", rec["box_err_px_pct"], " bins", rec["box_err_bins_pct"])
    agree = {}
    for name in r32:
        w32, i32 = r32[name]
        w16, i16 = r16[name]
        if i32 is not None and i32.dim() >= 2 and i32.shape[-1] == 1 or (i32 is not None and w32.dim() == 4 and w32.shape[-1] == 1):   # per-image ranked experts
            a = float((i32.reshape(i32.shape[0], -1) == i16.reshape(i16.shape[0], -1)).all(1).float().mean())
        else:                                                                          # per-token selection masks
            a = float(((w32 > 0) == (w16 > 0)).reshape(w32.shape[0], w32.shape[1], -1).all(1).float().mean())
        agree[name] = a
    rec["route_agreement"] = np.array(json.dumps(agree))
    print("[cfg5_l_ref16] routing agreement fp16 vs fp32 (share of images / tokens with identical expert sets):", {k: round(v, 4) for k, v in agree.items()})
    n32, n16 = cw_boxes(y32.numpy(), rcp["conf"], rcp["iou"], rcp["sigma"]), cw_boxes(y16.numpy(), rcp["conf"], rcp["iou"], rcp["sigma"])
    jac, cwd, boxd = [], [], []
    for (d32, i32, c32), (d16, i16, c16) in zip(n32, n16):
        s32, s16 = set(i32.tolist()), set(i16.tolist())
        jac.append(len(s32 & s16) / max(len(s32 | s16), 1))
        p16 = {int(a): j for j, a in enumerate(i16)}
        common = [(j, p16[int(a)]) for j, a in enumerate(i32) if int(a) in p16]
        if common:
            ja, jb = np.array(common).T
            cwd.append(np.abs(c32[ja] - c16[jb]).max(1))
            boxd.append(np.abs(d32[ja, :4] - d16[jb, :4]).max(1))
    rec["nms_jaccard"] = np.array(jac)
    rec["cw_box_err_px_pct"] = np.percentile(np.concatenate(cwd), PCT) if cwd else np.zeros(3)
    rec["nms_box_err_px_pct"] = np.percentile(np.concatenate(boxd), PCT) if boxd else np.zeros(3)
    print(f"[cfg5_l_ref16] NMS kept-set Jaccard per image {np.round(rec['nms_jaccard'], 4)}; common survivors: plain boxes px pct {rec['nms_box_err_px_pct']}, "
          f"Cluster-Weighted boxes px pct {rec['cw_box_err_px_pct']}")
    np.savez_compressed(HERE / "cfg5_l_ref16.npz", **rec)
    print("[cfg5_l_ref16] wrote", (HERE / "cfg5_l_ref16.npz").stat().st_size, "bytes")
