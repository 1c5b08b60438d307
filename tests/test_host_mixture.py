"""Config-5 rows on the CPU: the product's host orchestration of the mixture modules (``yolo_master_amd/nn/mixture.py``)
over the emulated libymk entry points (``tests/emu_ops.py``, see ``tests/test_host_emu.py`` for what that proves),
against golden vectors produced by the REAL reference modules (``tests/golden/make_golden_{moa,mot,gated,cfg5}.py``).

The kernels behind the config-5 entry points are not in libymk yet: this pins the dataflow (packing, channel padding,
buffer slicing, op order) that those kernels will be dropped into, module by module and for the whole model."""
import numpy as np
import pytest
import torch



def _load(golden_dir, fam, name):
    z = np.load(golden_dir / f"{fam}_{name}.npz")
    sd = {k: torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}
    return z, sd


def _prep(mod, sd):
    mod.load_state_dict(sd)
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3
    return mod.eval()


def _close(got, ref, what, rtol=2e-5):
    err = float((got - ref).abs().max())
    assert err <= rtol * max(1.0, float(ref.abs().max())), f"{what}: max |d| = {err:.3e} (|ref| max {float(ref.abs().max()):.3f})"


MOA_CASES = {"exact": {}, "blend": {}, "linear": {}, "kvcap": dict(regional_max_kv_tokens=64, shortcut=False),
             "hd21": dict(dim=128),     # BASELINE config 5 (L scale): head_dim 21, padded to 24 channels per head
             # sparse inference (moa/block.py:194-234): the global head skipped / only the group with the largest mean gate runs
             "sparse": dict(sparse_inference=True, sparse_inference_threshold=0.2),
             "sparse_one": dict(sparse_inference=True, sparse_inference_threshold=0.99)}


@pytest.mark.parametrize("name", list(MOA_CASES))
def test_moa_block_host_vs_reference(name, golden_dir, emu):
    from yolo_master_amd.nn.mixture import MoABlock

    z, sd = _load(golden_dir, "moa", name)
    kw = dict(MOA_CASES[name])
    m = _prep(MoABlock(kw.pop("dim", 48), num_heads=6, **kw), sd)
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    with torch.inference_mode():
        got = m(x)
    _close(got, y, f"moa_{name}")
    probs = m.last_route["weights"].permute(0, 3, 1, 2)          # [B, 3, H, W]
    assert float((probs - torch.from_numpy(z["router_probs"])[0]).abs().max()) <= 1e-5
    used = emu.CALLS
    if name.startswith("sparse"):
        act = [bool(a) for a in z["active"].tolist()]
        assert m.last_route["active"] == act and m.last_route["executed_groups"] == sum(act)
        assert abs(m.last_route["dropped_routing_mass"] - float(z["dropped_routing_mass"])) <= 1e-5
        assert used.get("window_attention", 0) == int(act[0]) and used.get("attention", 0) == int(act[1]) + int(act[2])   # skipped heads are not computed
        assert used["group_norm"] == 1 + sum(act) and used["weighted_sum"] == 1 and used["moa_sparse_gate"] == 1
        return
    assert used["window_attention"] == 1 and used["group_norm"] == 4 and used["weighted_sum"] == 1
    if name in ("linear", "kvcap"):
        assert used.get("linear_attention") == 1 and used["attention"] == 1      # regional only
    elif name == "blend":
        assert used.get("linear_attention") == 1 and used["attention"] == 2 and used["lerp"] == 1
    else:
        assert "linear_attention" not in used and used["attention"] == 2


def test_c2f_moa_host_vs_reference(golden_dir, emu):
    from yolo_master_amd.nn.mixture import C2fMoA

    z, sd = _load(golden_dir, "moa", "c2f")
    with pytest.warns(UserWarning, match="adjusted to 3"):      # the reference warns about the head count too
        m = _prep(C2fMoA(64, 96, n=2, num_heads=6), sd)
    with torch.inference_mode():
        got = m(torch.from_numpy(z["x"]))
    _close(got, torch.from_numpy(z["y"]), "moa_c2f")


def test_config5_entry_points_have_no_cpu_path():
    """Config-5 entry points refuse CPU tensors (`require_gpu`): no silent PyTorch path."""
    from yolo_master_amd import ops

    x = torch.zeros(1, 4, 4, 8)
    for call in (lambda: ops.group_norm(x, 2, None, None, 1e-5), lambda: ops.layer_norm(x, None, None, 1e-5),
                 lambda: ops.adaptive_avg_pool(x, 2, 2), lambda: ops.channel_stats(x), lambda: ops.channel_gate(x, x[:, 0, 0])):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


MOT_CASES = {"top2": {}, "shift": dict(window_shift=True, local_attn_window=7), "top1": dict(top_k=1), "dense": dict(top_k=3),
             "skip": {},
    "scene": dict(scene_aware_router=True), "scene3": dict(scene_aware_router=True, scene_hidden_dim=5, top_k=1),
    "image": dict(use_spatial_router=False), "image_scene": dict(use_spatial_router=False, scene_aware_router=True),
    "scene_bypass": dict(scene_aware_router=True, scene_inference_mode="bypass"),
             }


@pytest.mark.parametrize("name", list(MOT_CASES))
def test_mot_block_host_vs_reference(name, golden_dir, emu):
    from yolo_master_amd.nn.mixture import MoTBlock

    z, sd = _load(golden_dir, "mot", name)
    m = _prep(MoTBlock(48, num_heads=6, **MOT_CASES[name]), sd)
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    with torch.inference_mode():
        got = m(x)
    _close(got, y, f"mot_{name}")
    w = m.last_route["weights"].permute(0, 3, 1, 2)                               # [B, 3, H, W]
    rw, ridx = torch.from_numpy(z["router_w"])[0], torch.from_numpy(z["router_idx"])[0]   # [B, 3, H, W] / [B, k, H, W] (1 x 1 maps: image-level router)
    rw, ridx = rw.expand(-1, -1, *w.shape[2:]), ridx.expand(-1, -1, *w.shape[2:])
    assert float((w - rw).abs().max()) <= 1e-5
    if "scene_stats" in z.files:
        assert m.router.last_scene_applied and emu.CALLS["scene_bias"] == 1
        assert float((m.router.last_scene_stats - torch.from_numpy(z["scene_stats"])[0]).abs().max()) <= 1e-5
    else:
        assert not m.router.last_scene_applied and emu.CALLS.get("scene_bias", 0) == 0
    if name == "scene_bypass":
        assert m.router.last_scene_bypass_reason == "inference_policy_bypass"
    sel = torch.zeros_like(w, dtype=torch.bool).scatter_(1, ridx, True)
    assert torch.equal(w > 0, sel), "selected experts differ from the reference"
    active = m.last_route["active"]
    assert torch.equal(active.bool(), sel.flatten(2).any(2))
    if name == "skip":
        assert not bool(active[:, 2].any())
    if name == "shift":
        assert emu.CALLS["window_attention"] == 2                                 # local-window expert + shifted-window expert


def test_c2f_mot_host_vs_reference(golden_dir, emu):
    from yolo_master_amd.nn.mixture import C2fMoT

    z, sd = _load(golden_dir, "mot", "c2f")
    m = _prep(C2fMoT(64, 96, n=2, num_heads=6), sd)
    with torch.inference_mode():
        got = m(torch.from_numpy(z["x"]))
    _close(got, torch.from_numpy(z["y"]), "mot_c2f")
    assert emu.CALLS["deform_attention"] == 2 and emu.CALLS["layer_norm"] == 8


GATED_CASES = {"base": (64, {}), "small": (64, {}), "keep1": (64, {}), "e6k3": (96, dict(num_experts=6, top_k=3)),
               "e16": (64, dict(num_experts=16, top_k=2)), "mid": (64, {})}


@pytest.mark.parametrize("name", list(GATED_CASES))
def test_gated_moe_host_vs_reference(name, golden_dir, emu):
    from yolo_master_amd.nn.mixture import VisualEnhancedAdaptiveGateMoE

    C, kw = GATED_CASES[name]
    z, sd = _load(golden_dir, "gated", name)
    m = _prep(VisualEnhancedAdaptiveGateMoE(C, C, **kw), sd)
    assert m.expert_backend == ("shared_inverted" if name == "e16" else "low_rank_fused")
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    with torch.inference_mode():
        got = m(x)
    B = x.shape[0]
    r = m.last_route
    assert np.array_equal(r["indices"].numpy(), z["indices"].reshape(B, -1)), "routed experts differ from the reference"
    assert float(np.abs(r["weights"].reshape(B, -1).numpy() - z["weights"].reshape(B, -1)).max()) <= 1e-5
    _close(got, y, f"gated_{name}", rtol=5e-5)
    assert emu.CALLS["expert_conv"] == 1 and emu.CALLS["gated_route_decide"] == 1 and emu.CALLS["channel_shuffle_cat"] == 1


GATED2_CASES = ["opt_base", "opt_e16", "fus_base", "fus_small", "fus_e16", "fus_keep1", "mh_base", "mh_e16", "mh_h3",          # mh: MultiHeadRouterMoE (v0_13)
                "div_base", "div_e16", "div_k3", "div_keep1"]                                                       # div: DiversifiedExpertMoE (v0_14)


def run_gated2_case(name, golden_dir, dev="cpu", dtype=torch.float32, rtol=5e-5):
    """OptimalHybridGateMoE (v0_12) / GatedFusionMoE (v0_15) against the real reference's vectors; shared with the GPU test."""
    from yolo_master_amd.nn import mixture

    z, sd = _load(golden_dir, "gated2", name)
    cls = getattr(mixture, str(z["cls"]))
    kw = eval(str(z["kw"]), {"__builtins__": {}}, {"dict": dict})
    m = _prep(cls(128, 128, **kw), sd)
    assert m.expert_backend == ("diversified" if name.startswith("div") else "shared_inverted" if kw.get("num_experts", 4) > 8 else "fused")
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    if dev != "cpu":
        from yolo_master_amd.nn.modules import set_compute_dtype

        m = m.to(dev)
        set_compute_dtype(m, dtype)
    with torch.inference_mode():
        got = m(x.to(dev))
    B = x.shape[0]
    r = m.last_route
    assert np.array_equal(r["indices"].cpu().numpy(), z["indices"].reshape(B, -1)), "routed experts differ from the reference"
    assert float(np.abs(r["weights"].reshape(B, -1).cpu().numpy() - z["weights"].reshape(B, -1)).max()) <= 1e-5
    _close(got.float().cpu(), y, f"gated2_{name}", rtol=rtol)
    return m


@pytest.mark.parametrize("name", GATED2_CASES)
def test_gated_v12_v15_host_vs_reference(name, golden_dir, emu):
    run_gated2_case(name, golden_dir)
    assert emu.CALLS["expert_conv"] == 1 and emu.CALLS["gated_route_decide"] == 1 and emu.CALLS["channel_shuffle_cat"] == 1
    assert emu.CALLS["layer_norm"] == 1 and emu.CALLS.get("expert_dw3", 0) == (1 if name.startswith("div") else 0)


GATED3_CASES = ["agm", "agm_hooks", "agm_keep1", "fused", "hyb", "hyb_e16", "hyb2", "lowrank", "refined", "detail", "ctxref"]


def run_gated3_case(name, golden_dir, dev="cpu", dtype=torch.float32, rtol=5e-5):
    """The earlier generations of the gated family (AdaptiveGateMoE v0_4 ... ContextRefined, HybridAdaptiveGateMoEv2 v0_11;
    moe/gated.py:268-1700) against the real reference's vectors; shared with the GPU test."""
    from yolo_master_amd.nn import mixture

    z, sd = _load(golden_dir, "gated3", name)
    cls = getattr(mixture, str(z["cls"]))
    kw = eval(str(z["kw"]), {"__builtins__": {}}, {"dict": dict})
    m = _prep(cls(64, 64, **kw), sd)
    assert list(m.state_dict().keys()) == z["keys"].tolist(), "state_dict keys / order differ from the reference's"
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    if dev != "cpu":
        from yolo_master_amd.nn.modules import set_compute_dtype

        m = m.to(dev)
        set_compute_dtype(m, dtype)
    with torch.inference_mode():
        got = m(x.to(dev))
    B = x.shape[0]
    r = m.last_route
    assert np.array_equal(r["indices"].cpu().numpy(), z["indices"].reshape(B, -1)), "routed experts differ from the reference"
    assert float(np.abs(r["weights"].reshape(B, -1).cpu().numpy() - z["weights"].reshape(B, -1)).max()) <= 1e-5
    _close(got.float().cpu(), y, f"gated3_{name}", rtol=rtol)
    return m


@pytest.mark.parametrize("name", GATED3_CASES)
def test_gated_chain_host_vs_reference(name, golden_dir, emu):
    m = run_gated3_case(name, golden_dir)
    assert emu.CALLS["expert_conv"] == 1 and emu.CALLS["gated_route_decide"] == 1 and emu.CALLS["channel_shuffle_cat"] == 1
    assert emu.CALLS.get("layer_norm", 0) == (1 if name == "hyb2" else 0)
    assert m.expert_backend == {"agm": "shared_inverted", "agm_hooks": "shared_inverted", "agm_keep1": "shared_inverted", "fused": "fused",
                                "hyb": "fused", "hyb_e16": "shared_inverted", "hyb2": "fused"}.get(name, "low_rank_fused")


MODEL_FIXTURES = {"cfg5": "yolo-master-moa-mot-n.yaml", "v15": "yolo-master-v15-n.yaml",
                  "v04": None, "v06": None, "v01": None, "v03": None, "v08s": None, "uomoe": None}    # None: the reference YAML's dict as stored in the fixture (generations v0_4 / v0_6, n scale)


@pytest.mark.parametrize("tag", list(MODEL_FIXTURES))
def test_config5_model_host_vs_reference(tag, golden_dir, emu):
    """Whole detectors of the gated-MoE generations through the product's graph walk, against the real reference model's
    per-layer samples and routing decisions: config 5 (v0_10 moa-mot YAML: VisualEnhancedAdaptiveGateMoE backbone, C2fMoT /
    C2fMoA neck), the v0_15 YAML (GatedFusionMoE backbone on the v0 neck / head) and the v0_4 / v0_6 YAMLs (AdaptiveGateMoE /
    HybridAdaptiveGateMoE with 4, 8 and 16 experts: shared-inverted and fused backends)."""
    import json

    from tests.helpers import fill_by_name
    from yolo_master_amd import ops
    from yolo_master_amd.nn.tasks import DetectionModel

    z = np.load(golden_dir / f"fwd_{tag}.npz")
    cfg = json.loads(str(z["cfg"]))
    sd = fill_by_name(json.loads(str(z["spec"])), seed=5, gain=1.0)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    m = DetectionModel(MODEL_FIXTURES[tag] or cfg)
    keys = json.load(open(golden_dir / f"keys_{tag}.json"))   # the reference model's own state_dict (names, order, shapes): the drop-in contract
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(k, v) for k, v in keys.items()], "state_dict differs from the reference's"
    m.load_state_dict(sd)
    m.eval()
    taps = {}
    with torch.inference_mode():
        y, preds = m._predict_once(torch.from_numpy(z["x"]), taps=taps)
    # discrete decisions: routed experts of the three gated blocks, selected experts per token of every MoT block
    for key in [f for f in z.files if f.startswith("route::")]:
        name = key[len("route::"):]                       # e.g. model.5 / model.14.m.0
        mod = m
        for part in name.split("."):
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
        ref = z[key]
        r = mod.last_route
        if "indices" in r:                                # gated block: [B, k]
            assert np.array_equal(r["indices"].numpy().reshape(ref.shape).astype(np.int16), ref), key
        else:                                             # MoT block: top-k expert ids per token [B, k, H, W]
            w = r["weights"].permute(0, 3, 1, 2)
            sel = torch.zeros_like(w, dtype=torch.bool).scatter_(1, torch.from_numpy(ref.astype(np.int64)), True)
            assert torch.equal(w > 0, sel), key
    n = len(cfg["backbone"]) + len(cfg["head"])
    worst = 0.0
    for i in range(n - 1):
        t = taps[i]
        if not torch.is_tensor(t):
            t = t.materialise()
        got = ops.nhwc_to_nchw_f32(t).reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        ref = z[f"layer{i}_val"]
        err = float(np.abs(got - ref).max() / max(1.0, float(np.abs(ref).max())))
        worst = max(worst, err)
        assert err <= 1e-3, f"layer {i} ({(cfg['backbone'] + cfg['head'])[i][2]}): scaled max error {err:.3e}"
    # Detect output.  The fixture's name-seeded weights also randomise `dfl.conv.weight` (a frozen arange in every real
    # checkpoint, which is what the decode kernel evaluates in closed form), so the reference's boxes are re-derived
    # here from the product's raw box logits with the fixture's DFL weights; class scores are compared directly.
    import torch.nn.functional as F

    det = m.model[-1]
    B, A = y.shape[0], y.shape[2]
    raw = preds["raw"]
    boxes = torch.cat([b.reshape(B, -1, 4 * det.reg_max).transpose(1, 2) for b, _ in raw], -1)
    dist = F.conv2d(boxes.view(B, 4, det.reg_max, A).transpose(2, 1).softmax(1), sd[f"model.{n - 1}.dfl.conv.weight"]).view(B, 4, A)
    anc, strd = [], []
    for (b, _), st in zip(raw, det.stride.tolist()):
        hh, ww = b.shape[1:3]
        sy, sx = torch.meshgrid(torch.arange(hh, dtype=torch.float32) + 0.5, torch.arange(ww, dtype=torch.float32) + 0.5, indexing="ij")
        anc.append(torch.stack((sx, sy), -1).view(-1, 2))
        strd.append(torch.full((hh * ww, 1), float(st)))
    anc, strd = torch.cat(anc).t(), torch.cat(strd).t()
    x1y1, x2y2 = anc - dist[:, :2], anc + dist[:, 2:]
    yref = torch.cat([torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], 1) * strd, y[:, 4:]], 1)
    got = yref.reshape(-1)[torch.from_numpy(z["y_idx"].astype(np.int64))].numpy()
    np.testing.assert_allclose(got, z["y_val"], rtol=1e-3, atol=2e-2)
    print(f"config-5 host walk: worst scaled layer error {worst:.3e}")


def test_config5_model_bf16_operand_rules(golden_dir, emu):
    """bf16 compute dtype: every operand the host hands to a kernel satisfies the 16-byte channel-vector rules (asserted
    inside the emulation), routers / statistics / sampling coordinates stay fp32, and the result tracks fp32."""
    import json

    from tests.helpers import fill_by_name
    from yolo_master_amd.nn.tasks import DetectionModel

    z = np.load(golden_dir / "fwd_cfg5.npz")
    sd = fill_by_name(json.loads(str(z["spec"])), seed=5, gain=1.0)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    x = torch.from_numpy(z["x"])
    ys = {}
    for dt in (torch.float32, torch.bfloat16):
        m = DetectionModel("yolo-master-moa-mot-n.yaml")
        m.load_state_dict(sd)
        m.eval().set_compute_dtype(dt)
        with torch.inference_mode():
            ys[dt], _ = m._predict_once(x)
    assert bool(torch.isfinite(ys[torch.bfloat16]).all())
    d = (ys[torch.bfloat16][:, 4:] - ys[torch.float32][:, 4:]).abs()
    # name-seeded weights: uncalibrated scores around 0.5 and per-token top-k routing that flips under bf16 rounding, so
    # only a loose bound is meaningful here (the calibrated drift test of the detector lives in test_gpu_model.py)
    assert float(d.median()) < 5e-2, f"median class-score drift {float(d.median()):.3e}"


def test_config5_L_scale_host_vs_oracle(emu):
    """BASELINE config 5 is the moa-mot YAML at the L scale (width 1.0, depth 1.0: 51.7 M parameters; MoA head_dim 21,
    16-expert shared-inverted block, 256-wide MoT blocks with four repeats).  No reference fixture of that size is
    committed, so the product's host path (emulated entry points) is compared with the oracle — itself pinned bit-exact
    against the real reference on every module family and on the n-scale model — layer by layer, in fp32; the bf16 pass
    checks that every operand still satisfies the kernels' channel-vector rules at these widths."""
    import copy

    import yaml

    from oracle import model_ref
    from tests.helpers import fill_by_name
    from yolo_master_amd import ops
    from yolo_master_amd.nn.tasks import CFG_DIR, DetectionModel

    d = yaml.safe_load(open(CFG_DIR / "yolo-master-moa-mot.yaml"))
    d["scales"]["l"] = [1.0, 1.0, 512]
    d["scale"] = "l"
    m = DetectionModel(copy.deepcopy(d))
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 51.7) < 0.1
    assert m.model[17].m[0].local_head.head_dim == 21 and m.model[11].expert_backend == "shared_inverted"
    spec = {k: list(v.shape) for k, v in m.state_dict().items() if v.is_floating_point() and v.dim() > 0 and not k.endswith("_rf_matrix")}
    full = dict(m.state_dict())
    full.update(fill_by_name(spec, seed=9, gain=0.7))     # gain < 1: keeps the 16-block residual chains well conditioned
    m.load_state_dict(full)
    m.eval()
    x = torch.rand(1, 3, 160, 128, generator=torch.Generator().manual_seed(1))
    taps, otaps = {}, {}
    with torch.inference_mode():
        y, _ = m._predict_once(x, taps=taps)
        oy, _, _ = model_ref.forward(copy.deepcopy(d), dict(m.state_dict()), x, fused=False, taps=otaps)
    for i in range(len(m.model) - 1):
        t = taps[i] if torch.is_tensor(taps[i]) else taps[i].materialise()
        ref = otaps[i]
        err = float((ops.nhwc_to_nchw_f32(t) - ref).abs().max() / max(1.0, float(ref.abs().max())))
        assert err <= 1e-4, f"layer {i} ({type(m.model[i]).__name__}): scaled max error {err:.3e}"
    assert float((y[:, 4:] - oy[:, 4:]).abs().max()) <= 1e-5
    # the 32-wide heads of the L-scale MoT blocks (whole-map attention, the dominant cost at 1280 px) run on the MFMA
    # area-attention kernel of the A2C2f blocks: 16 A2C2f calls + 3 layers x 4 blocks
    assert emu.CALLS["area_attn"] == 16 + 12 and emu.CALLS["attention"] == 4   # MoA: regional + exact global, two blocks
    m.set_compute_dtype(torch.bfloat16)
    with torch.inference_mode():
        yb, _ = m._predict_once(x)
    assert bool(torch.isfinite(yb).all())
