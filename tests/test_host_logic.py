"""Host-side logic on CPU: YAML parsing / drop-in state_dict contract, packing algebra, synthetic weights,
and the 'no CPU fallback' rule."""
import json
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("scale", ["n", "s", "l"])
def test_state_dict_contract_matches_reference(scale):
    from yolo_master_amd.nn.tasks import DetectionModel

    m = DetectionModel(f"yolo-master-{scale}.yaml")
    keys = json.load(open(GOLD / f"keys_{scale}.json"))   # dumped from the real reference model
    sd = m.state_dict()
    assert list(sd.keys()) == list(keys.keys())
    assert all(list(sd[k].shape) == v for k, v in keys.items())
    assert m.stride.tolist() == [8.0, 16.0, 32.0]
    assert m.save == [6, 9, 12, 15, 18, 21, 24]
    assert all(bn.eps == 1e-3 for bn in m.modules() if isinstance(bn, torch.nn.BatchNorm2d))


def test_all_scales_build():
    from yolo_master_amd.nn.tasks import DetectionModel

    n_params = {}
    for s in "nsmlx":
        m = DetectionModel(f"yolo-master-{s}.yaml")
        n_params[s] = sum(p.numel() for p in m.parameters())
    assert n_params["n"] < n_params["s"] < n_params["m"] < n_params["l"] < n_params["x"]


def test_routed_module_protocol_surface():
    from yolo_master_amd.nn.modules import ES_MOE

    m = ES_MOE(64, 64)
    assert (m.num_experts, m.top_k) == (4, 2)
    assert [e.conv.depthwise.kernel_size[0] for e in m.experts] == [3, 5, 7, 9]
    assert ES_MOE(64, 64, num_experts=3).experts[2].conv.depthwise.kernel_size[0] == 7
    assert ES_MOE(64, 64, num_experts=8, max_kernel_size=9).experts[7].conv.depthwise.kernel_size[0] == 9
    caps = m.export_capabilities()
    assert caps["sparse_dispatch"] and caps["routing_kind"] == "moe"
    m.set_top_k(None)
    assert not m._eager_sparse_enabled()
    assert "load_balancing_loss" not in m.state_dict() and "expert_usage_counts" not in m.state_dict()
    for bad in (dict(num_experts=0), dict(top_k=5), dict(dynamic_threshold=1.5), dict(max_kernel_size=1), dict(reduction=0)):
        with pytest.raises(ValueError):
            ES_MOE(64, 64, **bad)
    with pytest.raises(ValueError):
        ES_MOE(64, 64, expert_kernel_sizes=[3, 5])


def test_fold_and_pack_algebra():
    from yolo_master_amd import ops

    g = torch.Generator().manual_seed(0)
    w = torch.randn(8, 16, 3, 3, generator=g)
    bn = torch.nn.BatchNorm2d(8).eval()
    bn.eps = 1e-3
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
        wf, bf = ops.fold_bn(w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        x = torch.randn(2, 16, 9, 9, generator=g)
        ref = bn(F.conv2d(x, w, None, 1, 1))
        assert torch.allclose(F.conv2d(x, wf, bf, 1, 1), ref, atol=1e-5)
    p = ops.pack_conv_weight(w, torch.float32)
    assert p.shape == (8, 192) and ops.kpad(16 * 9) == 192
    assert torch.equal(p[:, :144].reshape(8, 3, 3, 16), w.permute(0, 2, 3, 1)) and float(p[:, 144:].abs().max()) == 0
    d = ops.pack_dw_weight(torch.arange(2 * 9.0).reshape(2, 1, 3, 3), torch.float32)
    assert d.shape == (9, 2) and d[4].tolist() == [4.0, 13.0]


def test_synthetic_weights_are_deterministic_and_calibrated():
    from yolo_master_amd.nn.tasks import DetectionModel
    from yolo_master_amd.weights import synth_input, synth_state_dict

    t = DetectionModel("yolo-master-n.yaml").state_dict()
    a, b = synth_state_dict(t, seed=0), synth_state_dict(t, seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    raw = synth_state_dict(t, seed=0, calib=None)
    assert not torch.equal(a["model.0.bn.running_var"], raw["model.0.bn.running_var"]), "bn_calib_n.npz not applied"
    assert torch.equal(a["model.25.dfl.conv.weight"].reshape(-1), torch.arange(16.0))
    x = synth_input(3, 64, 96, seed=5)
    assert x.shape == (3, 3, 64, 96) and 0 <= float(x.min()) and float(x.max()) <= 1
    assert torch.equal(x, synth_input(3, 64, 96, seed=5))


def test_no_cpu_fallback():
    """The product path must fail loudly instead of computing anything on the CPU."""
    from yolo_master_amd.nn.modules import Conv
    from yolo_master_amd.nn.tasks import DetectionModel

    m = DetectionModel("yolo-master-n.yaml").eval()
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="MI355X"):
        Conv(16, 16, 3).eval()(torch.zeros(1, 16, 8, 8))
    with pytest.raises(RuntimeError, match="eval-mode"):
        DetectionModel("yolo-master-n.yaml").train()(torch.zeros(1, 3, 64, 64))


def test_product_never_imports_oracle():
    """Neither the oracle nor the test-side emulations (tests/emu_ops.py: torch restatement of the C-ABI contracts;
    tests/hostemu: CPU lane emulator) are reachable from the product: no import statement mentions them."""
    import re

    root = Path(__file__).resolve().parent.parent / "yolo_master_amd"
    for f in root.rglob("*.py"):
        txt = f.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, f
        for line in txt.splitlines():
            if re.match(r"\s*(import|from)\s", line):
                assert "tests" not in line.split() and "emu_ops" not in line and "hostemu" not in line, f"{f}: {line}"


def test_lx_scale_host_logic():
    """l/x scales: A2C2f carries the gamma-residual and mlp_ratio 1.2 (tasks.py:2156-2159); the MLP hidden width
    int(1.2*dim) is zero-padded at pack time so that every conv sees 16-byte channel vectors (and whole 64-channel k-steps)."""
    from yolo_master_amd.nn.modules import A2C2f, ABlock
    from yolo_master_amd.nn.tasks import DetectionModel

    m = DetectionModel("yolo-master-l.yaml").eval()
    blocks = [x for x in m.model if isinstance(x, A2C2f)]
    assert blocks and all(b.gamma is not None and b.gamma.shape == (b.cv2.conv.out_channels,) for b in blocks)
    ab = next(x for x in m.modules() if isinstance(x, ABlock))
    hidden = ab.mlp[0].conv.out_channels
    assert hidden == int(ab.mlp[0].conv.in_channels * 1.2) and hidden % 8 != 0
    hp = (hidden + 63) // 64 * 64    # 307 -> 320: a multiple of 64 where that costs <= 1/8 more arithmetic (both convolutions on the LDS-DMA core), else of 8
    assert hp - hidden <= hidden // 8 and ab.mlp[0].pad_cout_to == hp == ab.mlp[1].pad_cin_to
    with torch.no_grad():
        p0, p1 = ab.mlp[0]._pack(torch.float32, "cpu"), ab.mlp[1]._pack(torch.float32, "cpu")
    assert p0["w"].shape[0] == hp and p0["b"].shape[0] == hp
    assert float(p0["w"][hidden:].abs().max()) == 0.0 and float(p0["b"][hidden:].abs().max()) == 0.0   # SiLU(0) = 0
    k1 = p1["w"].shape[1]
    assert k1 >= hp and float(p1["w"][:, hidden:].abs().max()) == 0.0                                   # zero columns
    # the reference's key set for the L scale (dumped from the real model) is reproduced
    import json
    keys = json.load(open(Path(__file__).parent / "golden" / "keys_l.json"))
    assert list(m.state_dict().keys()) == list(keys.keys())


def test_conv_kernel_names_match_the_committed_profiles(tmp_path, monkeypatch):
    """The timer tags conv calls with the demangled kernel name; bench.py's roofline.traffic lookup and the judge's
    cross-check against profiles/*_kernel_stats.txt rely on these strings being what rocprofv3 prints.  A committed counter summary
    is only quoted for the kernel sources it was collected on (`csrc_sha16`): a profile of other sources must yield None."""
    import json
    import shutil

    import bench
    from yolo_master_amd import ops
    from yolo_master_amd.build import source_hash

    latest = sorted((Path(__file__).parent.parent / "profiles").glob("*_pmc_FETCH_SIZE.json"))[-1]
    prof = json.load(open(latest))["kernels"]
    bf = torch.bfloat16
    names = [
        ops.conv_kernel_name(0, bf, 256, 128, 1, 256, True),              # tiled 1x1, 128x128 tile
        ops.conv_kernel_name(0, bf, 64, 64, 1, 64, False),                # tiled 1x1, 64x256 tile
        ops.conv_kernel_name(3 | (2 << 8) | (128 << 16), bf, 256, 256, 3, 2304, False),  # LDS-DMA tiled 3x3, 128 couts, two stages, 128-pixel tiles
        ops.conv_kernel_name(3 | (2 << 8) | (128 << 16), bf, 128, 64, 3, 1152, False),  # ... 64 couts
        ops.conv_kernel_name(1, bf, 96, 128, 1, 128, False),              # streaming 1x1, 2 K groups
        ops.conv_kernel_name(2, bf, 32, 32, 3, 320, True),                # spatial tile 3x3 with residual prefetch
    ]
    present = [n for n in names if n in prof]
    assert len(present) >= 3, f"none of the expected kernel names is in the committed profile {latest.name}: {names}"
    assert ops.conv_kernel_name(0, torch.float32, 16, 8, 3, 192, False) == "conv_igemm_kernel<float, 16, 256, 1, 4, 3, false>"
    assert ops.conv_kernel_name(0, torch.float16, 64, 64, 1, 64, False) == ops.conv_kernel_name(0, bf, 64, 64, 1, 64, False)   # fp16 build: same names
    # staleness guard: the same summaries under another source hash are ignored; stamped with the current one they are used
    monkeypatch.setattr(bench, "PROFILES", tmp_path)
    for suffix in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE"):
        d = json.load(open(str(latest).replace("pmc_FETCH_SIZE", suffix)))
        d["csrc_sha16"] = "0" * 16
        json.dump(d, open(tmp_path / f"t00_{suffix}.json", "w"))
    assert bench.pmc_traffic(present[0]) is None, "a profile of other kernel sources must not be quoted"
    for suffix in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE"):
        d = json.load(open(tmp_path / f"t00_{suffix}.json"))
        d["csrc_sha16"] = source_hash()
        json.dump(d, open(tmp_path / f"t01_{suffix}.json", "w"))
    assert bench.pmc_traffic(present[0]) > 0
    assert bench.pmc_traffic("moe_dw") > 0 and bench.pmc_traffic("nms") is None   # prefix families / multi-kernel ops



def test_config5_boundary_modules_keep_the_reference_contract():
    """The v0_10 moa-mot model builds from the model YAML and its state_dict has the reference's 1200 keys, in the
    reference's order and shapes (dumped from the real model by tests/golden/make_golden_cfg5.py); the fixed
    random-feature bases of the MoA global heads equal the reference's; the mixture modules have no CPU path."""
    import warnings

    import numpy as np

    from yolo_master_amd.nn.mixture import C2fMoA, C2fMoT, MoABlock, VisualEnhancedAdaptiveGateMoE
    from yolo_master_amd.nn.tasks import DetectionModel

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DetectionModel("yolo-master-moa-mot-n.yaml").eval()
    keys = json.load(open(GOLD / "keys_cfg5.json"))
    sd = m.state_dict()
    assert list(sd.keys()) == list(keys.keys())
    assert all(list(sd[k].shape) == v for k, v in keys.items())
    z = np.load(GOLD / "fwd_cfg5.npz")
    rf = [k for k in z.files if k.endswith("_rf_matrix")]
    assert rf and all(np.allclose(sd[k[len("fixed::"):]].numpy(), z[k], atol=1e-6) for k in rf)
    assert float(sd["model.20.m.0.router.temperature"]) == pytest.approx(0.8)
    kinds = [type(x) for x in m.model]
    assert kinds.count(VisualEnhancedAdaptiveGateMoE) == 3 and kinds.count(C2fMoT) == 3 and kinds.count(C2fMoA) == 1
    assert m.model[11].expert_backend == "shared_inverted" and m.model[5].expert_backend == "low_rank_fused"
    for mod in (m.model[5], m.model[14], m.model[17], MoABlock(48, 6).eval()):
        with pytest.raises(RuntimeError, match="MI355X"):      # CPU tensors are refused like everywhere else
            mod(torch.zeros(1, mod.cv1.conv.in_channels if hasattr(mod, "cv1") else 48 if isinstance(mod, MoABlock) else 128, 8, 8))


def test_shared_expert_pool_semantics():
    """SharedExpertMoE (moe/shared_expert_moe.py:85-120): blocks with one pool_id alias one expert group (first = owner), a signature mismatch
    raises ValueError, the registry is per model (parse_model resets it), and the shared tensors appear under every member's prefix."""
    import pytest as _pt

    from yolo_master_amd.nn.mixture import SharedExpertMoE
    from yolo_master_amd.nn.tasks import DetectionModel

    SharedExpertMoE.reset_shared_pools()
    a = SharedExpertMoE(64, 64, 4, 2, pool_id="p")
    b = SharedExpertMoE(64, 64, 4, 2, pool_id="p")
    c = SharedExpertMoE(64, 64, 4, 2, pool_id="q")
    assert a.fused_experts is b.fused_experts and a.fused_experts is not c.fused_experts
    assert a.get_pool_info()["is_owner"] and not b.get_pool_info()["is_owner"] and c.get_pool_info()["is_owner"]
    with _pt.raises(ValueError, match="parameter mismatch"):
        SharedExpertMoE(128, 128, 4, 2, pool_id="p")          # other dynamic width
    with _pt.raises(ValueError, match="parameter mismatch"):
        SharedExpertMoE(64, 64, 4, 1, pool_id="p")            # other top_k
    SharedExpertMoE.reset_shared_pools()
    assert SharedExpertMoE(64, 64, 4, 2, pool_id="p").get_pool_info()["is_owner"]
    cfg = {"nc": 3, "scales": {"n": [1.0, 1.0, 1024]}, "scale": "n",
           "backbone": [[-1, 1, "Conv", [64, 3, 2]], [-1, 1, "SharedExpertMoE", [64, 4, 2, 0.5, 8, 1.2, 0.5, 1.0, 1.0, 0.01, 8, 2, 0.5, "pp"]],
                        [-1, 1, "SharedExpertMoE", [64, 4, 2, 0.5, 8, 1.2, 0.5, 1.0, 1.0, 0.01, 8, 2, 0.5, "pp"]]],
           "head": [[[2], 1, "Detect", ["nc"]]]}
    m1, m2 = DetectionModel(cfg), DetectionModel(cfg)
    assert m1.model[1].fused_experts is m1.model[2].fused_experts
    assert m1.model[1].fused_experts is not m2.model[1].fused_experts, "pools must not leak between models"
    sd = m1.state_dict()
    k1 = [k for k in sd if k.startswith("model.1.fused_experts.")]
    assert k1 and all(("model.2." + k[len("model.1."):]) in sd and sd["model.2." + k[len("model.1."):]].data_ptr() == sd[k].data_ptr() for k in k1)
