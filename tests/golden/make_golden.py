"""Generate the golden fixtures in this directory FROM THE REAL REFERENCE (/root/reference).

Run in the build container only (the GPU box has no reference checkout):

    python tests/golden/make_golden.py

For every case the reference's own code is executed on CPU fp32 —
``ultralytics.nn.tasks.DetectionModel(yaml).eval().fuse()`` forward and
``ultralytics.utils.nms.non_max_suppression`` (pure-torch TorchNMS path) — on the seeded
synthetic weights/inputs of yolo_master_amd/weights.py, and compact outputs are stored as .npz:

  fwd_<case>.npz   sampled per-layer activations, sampled/full y, NMS detections + kept anchor indices,
                   ES_MOE routing weights (from the reference's router module) — model cases
  nms_<case>.npz   y + reference NMS results for synthetic prediction tensors (single/multi-label,
                   agnostic, caps, empty)
  keys_<scale>.json  the reference state_dict's key -> shape map (drop-in contract)

The script also cross-checks the oracle restatement (oracle/model_ref.py, oracle/nms_ref.py) against the
reference while it has both at hand and prints the verdict.
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from oracle import model_ref, nms_ref, refboot  # noqa: E402

refboot.boot()
from ultralytics.nn.tasks import DetectionModel as RefModel  # noqa: E402
from ultralytics.utils.nms import non_max_suppression as ref_nms  # noqa: E402

from yolo_master_amd.nn.tasks import yaml_model_load  # noqa: E402
from yolo_master_amd.weights import CFG_DIR, synth_input, synth_state_dict  # noqa: E402

assert "torchvision" not in sys.modules, "the reference must take its pure-torch TorchNMS path"
REF_YAML = "/root/reference/ultralytics/cfg/models/master/v0/det/yolo-master-{}.yaml"
NSAMP_LAYER, NSAMP_Y = 768, 32768


def sample_idx(numel, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (min(n, numel),), generator=g)


def has_ties(y, conf):
    """True per image if two candidates share a score (the reference's unstable argsort makes their order,
    hence possibly the keep set, implementation-defined)."""
    out = []
    for b in range(y.shape[0]):
        c = y[b, 4:].amax(0)
        c = c[c > conf]
        out.append(bool(len(torch.unique(c)) != len(c)))
    return out


def model_case(name, scale, B, H, W, seed, conf=0.25, iou=0.7, full_y=False, calib="auto", n64=None, stable=False):
    """calib: "auto" = the committed BatchNorm calibration (chaotic at 640 x 640, see tools/make_conditioned.py), or the
    name of a cfg/*.npz override set ("cond_n.npz": the well-conditioned weights of the BASELINE-size fixtures).
    n64: number of leading images evaluated in fp64 (all when None).  stable: also record, per image, whether the
    reference's OWN discrete decisions survive an evaluation-order perturbation (BN folded vs unfolded, and fp64 on the
    first n64 images): parity tests demand bit-exact kept indices on exactly those images."""
    ref = RefModel(REF_YAML.format(scale), ch=3, nc=80, verbose=False)
    sd = synth_state_dict(ref.state_dict(), seed=0, calib=calib if calib == "auto" else str(CFG_DIR / calib))
    ref.load_state_dict(sd)
    ref.eval()
    x = synth_input(B, H, W, seed=seed)
    taps, routes = {}, {}
    for m in ref.model:
        m.register_forward_hook(lambda mod, i, o, idx=m.i: taps.__setitem__(idx, o))
        if type(m).__name__ == "ES_MOE":
            m.routing.register_forward_hook(lambda mod, i, o, idx=m.i: routes.__setitem__(idx, o[:, :, 0, 0].clone()))
    with torch.inference_mode():
        ref.fuse(verbose=False)
        y, preds = ref(x)
        dets, keepi = ref_nms(y.clone(), conf, iou, return_idxs=True)
        dets_ml, keepi_ml = ref_nms(y.clone(), 0.05, 0.6, multi_label=True, max_det=100, max_time_img=10.0,
                                    return_idxs=True)
    # cross-check the oracle restatement
    cfg = yaml_model_load(f"yolo-master-{scale}.yaml")
    otaps, info = {}, {}
    with torch.inference_mode():
        oy, oboxes, oscores = model_ref.forward(cfg, sd, x, taps=otaps, moe_info=info)
    # fp64 evaluation of the same graph (validated oracle, unfused = the mathematically exact composition):
    # |reference_fp32 - fp64| is the reference's own round-off noise floor, stored per layer so that parity
    # tests can bound the HIP path by "as close to the exact result as the reference itself is".
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    taps64 = {}
    n64 = B if n64 is None else min(n64, B)
    with torch.inference_mode():
        y64, _, _ = model_ref.forward(cfg, sd64, x[:n64].double(), fused=False, taps=taps64)
    exact = all(torch.equal(taps[i], otaps[i]) for i in range(len(ref.model) - 1)) and torch.equal(y, oy)
    print(f"[{name}] oracle forward bit-exact vs reference: {exact}; max|dy| = {(y - oy).abs().max().item():.3e}")
    for i, rw in routes.items():
        assert torch.equal(rw, info[f"model.{i}"]["route_w"]), f"routing weights differ at layer {i}"
    ties = has_ties(y, conf)
    onms = nms_ref.non_max_suppression(y.numpy(), conf, iou, return_idxs=True)
    for b in range(B):
        same = np.array_equal(onms[1][b], keepi[b].numpy().reshape(-1)) and np.array_equal(onms[0][b], dets[b].numpy())
        print(f"[{name}] img {b}: cands={(y[b, 4:].amax(0) > conf).sum().item()} kept={len(keepi[b])} "
              f"ties={ties[b]} numpy-oracle NMS == reference: {same}")
        assert same or ties[b], "oracle NMS differs from the reference on a tie-free image"
    rec = {"B": B, "H": H, "W": W, "seed": seed, "conf": conf, "iou": iou, "scale": ord(scale), "n64": n64}
    if calib != "auto":
        rec["calib"] = np.array(calib)
    for i in range(len(ref.model) - 1):
        idx = sample_idx(taps[i].numel(), NSAMP_LAYER, 1000 + i)
        rec[f"layer{i}_idx"] = idx.numpy().astype(np.int32)
        rec[f"layer{i}_val"] = taps[i].reshape(-1)[idx].numpy()
        rec[f"layer{i}_shape"] = np.array(taps[i].shape)
        if n64 == B:
            rec[f"layer{i}_val64"] = taps64[i].reshape(-1)[idx].numpy()
        rec[f"layer{i}_noise"] = np.float64((taps[i][:n64].double() - taps64[i]).abs().max().item())
    if full_y:
        rec["y"] = y.numpy()
    idx = sample_idx(y.numel(), NSAMP_Y, 7)
    rec["y_idx"], rec["y_val"], rec["y_shape"] = idx.numpy().astype(np.int32), y.reshape(-1)[idx].numpy(), np.array(y.shape)
    if n64 == B:
        rec["y_val64"] = y64.reshape(-1)[idx].numpy()
    rec["y_noise_box"] = np.float64((y[:n64, :4].double() - y64[:, :4]).abs().max().item())
    rec["y_noise_cls"] = np.float64((y[:n64, 4:].double() - y64[:, 4:]).abs().max().item())
    if stable:
        with torch.inference_mode():
            info_u = {}
            yu, _, _ = model_ref.forward(cfg, sd, x, fused=False, moe_info=info_u)
        ku = nms_ref.non_max_suppression(yu.numpy(), conf, iou, return_idxs=True)[1]
        k64 = nms_ref.non_max_suppression(y64.float().numpy(), conf, iou, return_idxs=True)[1]
        st = [np.array_equal(ku[b], keepi[b].numpy().reshape(-1)) and (b >= n64 or np.array_equal(k64[b], keepi[b].numpy().reshape(-1)))
              for b in range(B)]
        rst = [all(torch.equal(info[f"model.{i}"]["retained"][b], info_u[f"model.{i}"]["retained"][b]) for i in routes) for b in range(B)]
        rec["stable"], rec["route_stable"] = np.array(st), np.array(rst)
        rec["perturb_box"] = np.float64((y[:, :4] - yu[:, :4]).abs().max().item())
        rec["perturb_cls"] = np.float64((y[:, 4:] - yu[:, 4:]).abs().max().item())
        print(f"[{name}] decisions stable under BN-fold / fp64 perturbation (|dy| {rec['perturb_box']:.2e} px, {rec['perturb_cls']:.2e}): "
              f"NMS {sum(st)}/{B} images, routing {sum(rst)}/{B}")
    print(f"[{name}] reference fp32 noise floor vs fp64: boxes {rec['y_noise_box']:.3e} px, scores {rec['y_noise_cls']:.3e}")
    for i, rw in routes.items():
        rec[f"route{i}_route_w"] = rw.numpy()
        rec[f"route{i}_gate_w"] = info[f"model.{i}"]["gate_w"].numpy()
        rec[f"route{i}_retained"] = info[f"model.{i}"]["retained"].numpy()
        rec[f"route{i}_logits"] = info[f"model.{i}"]["logits"].numpy()
        # eval-time state the reference module holds after this forward (modules.py:706-741)
        rec[f"route{i}_usage"] = ref.model[i].expert_usage_counts.numpy().copy()
        rec[f"route{i}_lbloss"] = ref.model[i].load_balancing_loss.numpy().copy()
        assert torch.equal(ref.model[i].expert_usage_counts, info[f"model.{i}"]["usage"])
        assert torch.equal(ref.model[i].load_balancing_loss, info[f"model.{i}"]["lb_loss"])
    rec["ties"] = np.array(ties)
    for b in range(B):
        rec[f"nms{b}_dets"], rec[f"nms{b}_idx"] = dets[b].numpy(), keepi[b].numpy().reshape(-1).astype(np.int64)
        rec[f"nmsml{b}_dets"], rec[f"nmsml{b}_idx"] = dets_ml[b].numpy(), keepi_ml[b].numpy().reshape(-1).astype(np.int64)
    np.savez_compressed(HERE / f"fwd_{name}.npz", **rec)
    json.dump({k: list(v.shape) for k, v in RefModel(REF_YAML.format(scale), ch=3, nc=80, verbose=False).state_dict().items()},
              open(HERE / f"keys_{scale}.json", "w"))


def synth_pred(B, nc, A, seed, logit_mean=-4.0, logit_std=1.5, quant=None):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, 2, A, generator=g) * 600 + 20
    wh = torch.rand(B, 2, A, generator=g) * 150 + 8
    cls = torch.sigmoid(torch.randn(B, nc, A, generator=g) * logit_std + logit_mean)
    if quant:
        cls = (cls * quant).round() / quant
    return torch.cat([xy, wh, cls], 1)


def nms_case(name, make_y, **kw):
    """make_y(seed) -> y; seeds are tried in order until no image has tied candidate scores, because the
    reference's unstable argsort leaves the order of equal scores implementation-defined."""
    conf = kw.get("conf_thres", 0.25)
    for seed in range(100, 200):
        y = make_y(seed)
        ml = kw.get("multi_label", False)
        tied = False
        mi = 4 + (kw.get("nc") or y.shape[1] - 4)     # class rows end here; rows behind them are carried (Segment mask coefficients)
        for b in range(y.shape[0]):
            c = y[b, 4:mi][y[b, 4:mi] > conf] if ml else y[b, 4:mi].amax(0)[y[b, 4:mi].amax(0) > conf]
            tied |= len(torch.unique(c)) != len(c)
        if not tied:
            break
    with torch.inference_mode():
        dets, keepi = ref_nms(y.clone(), return_idxs=True, max_time_img=10.0, **kw)
    args = dict(conf_thres=kw.get("conf_thres", 0.25), iou_thres=kw.get("iou_thres", 0.45),
                multi_label=kw.get("multi_label", False), agnostic=kw.get("agnostic", False),
                max_det=kw.get("max_det", 300), max_nms=kw.get("max_nms", 30000))
    o = nms_ref.non_max_suppression(y.numpy(), return_idxs=True, classes=kw.get("classes"), nc=kw.get("nc", 0), **args)
    ok = all(np.array_equal(o[1][b], keepi[b].numpy().reshape(-1)) and np.array_equal(o[0][b], dets[b].numpy())
             for b in range(y.shape[0]))
    print(f"[nms_{name}] seed={seed} kept/img={[len(k) for k in keepi]} numpy-oracle == reference: {ok}")
    assert ok
    rec = {"y": y.numpy(), **{f"arg_{k}": np.array(v) for k, v in args.items()}}
    if kw.get("classes") is not None:
        rec["arg_classes"] = np.array(kw["classes"])
    if kw.get("nc"):
        rec["arg_nc"] = np.array(kw["nc"])
    for b in range(y.shape[0]):
        rec[f"dets{b}"], rec[f"idx{b}"] = dets[b].numpy(), keepi[b].numpy().reshape(-1).astype(np.int64)
    np.savez_compressed(HERE / f"nms_{name}.npz", **rec)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "baseline":   # BASELINE configs 2 and 3 at their full size, conditioned weights
        model_case("n640_b32", "n", 32, 640, 640, seed=1, conf=0.5, calib="cond_n.npz", n64=2, stable=True)
        model_case("s640_b64", "s", 64, 640, 640, seed=1, conf=0.5, calib="cond_s.npz", n64=1, stable=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "classes":    # the `classes=` filter (utils/nms.py:63,131-136), single- and multi-label
        nms_case("classes", lambda s: synth_pred(3, 20, 1500, s, -3.0), conf_thres=0.25, iou_thres=0.7, classes=[1, 7, 19, 33])
        nms_case("classes_multi", lambda s: synth_pred(2, 12, 800, s, -2.5), conf_thres=0.05, iou_thres=0.6, multi_label=True,
                 classes=[0, 5])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "seg":   # nc + extra rows: the call of models/yolo/segment/predict.py -> detect/predict.py:54-65
        def seg_pred(B, nc, nm, A, seed, mean):      # (`nc=len(names)`, utils/nms.py:76-81,117,122,127): 32 mask coefficients ride along
            g = torch.Generator().manual_seed(seed + 1000)
            return torch.cat([synth_pred(B, nc, A, seed, mean), torch.randn(B, nm, A, generator=g)], 1)
        nms_case("seg", lambda s: seg_pred(2, 20, 32, 900, s, -3.5), conf_thres=0.25, iou_thres=0.7, nc=20)
        nms_case("seg_multi", lambda s: seg_pred(2, 12, 32, 800, s, -2.5), conf_thres=0.05, iou_thres=0.6, multi_label=True, nc=12,
                 classes=[0, 5, 7])
        nms_case("seg_caps", lambda s: seg_pred(2, 1, 8, 2000, s, -1.0), conf_thres=0.1, iou_thres=0.7, max_det=50, max_nms=500, nc=1,
                 agnostic=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "l":   # only the L-scale case (C3k blocks, gamma-residual A2C2f, mlp 1.2)
        model_case("l_tiny", "l", 1, 64, 64, seed=5, conf=0.002, full_y=True)
        sys.exit(0)
    model_case("n640", "n", 4, 640, 640, seed=1)
    model_case("n_ragged", "n", 3, 96, 160, seed=2, conf=0.002, full_y=True)
    model_case("n_tiny", "n", 1, 64, 64, seed=3, conf=0.002, full_y=True)
    model_case("s_small", "s", 2, 128, 128, seed=4, conf=0.002, full_y=True)
    model_case("l_tiny", "l", 1, 64, 64, seed=5, conf=0.002, full_y=True)
    nms_case("single", lambda s: synth_pred(3, 20, 1500, s, -3.0), conf_thres=0.25, iou_thres=0.7)
    nms_case("multi", lambda s: synth_pred(2, 12, 800, s, -2.5), conf_thres=0.05, iou_thres=0.6, multi_label=True)
    nms_case("agnostic", lambda s: synth_pred(2, 20, 1200, s, -3.0), conf_thres=0.25, iou_thres=0.45, agnostic=True)
    nms_case("caps", lambda s: synth_pred(2, 20, 2000, s, -2.0), conf_thres=0.1, iou_thres=0.7, max_det=50,
             max_nms=500)
    nms_case("empty", lambda s: synth_pred(2, 20, 300, s, -9.0), conf_thres=0.25, iou_thres=0.7)
    nms_case("one", lambda s: torch.cat([synth_pred(1, 20, 64, s, -9.0)[:, :, :63],
                                         torch.tensor([100., 100., 50., 40.] + [0.9] + [0.0] * 19).view(1, 24, 1)], 2),
             conf_thres=0.25, iou_thres=0.7)
