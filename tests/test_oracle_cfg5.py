"""Config 5 at model level (`v0_10/det/yolo-master-moa-mot-n.yaml`: VisualEnhancedAdaptiveGateMoE backbone, C2fMoT / C2fMoA
neck, Detect): the oracle's YAML walk with gated_ref / moa_ref / mot_ref against golden vectors produced by the REAL
reference model (tests/golden/make_golden_cfg5.py).  CPU only; the reference is not needed at run time."""
import json

import numpy as np
import torch

from tests.helpers import fill_by_name


def test_config5_model_oracle_reproduces_reference(golden_dir):
    from oracle import model_ref

    z = np.load(golden_dir / "fwd_cfg5.npz")
    cfg = json.loads(str(z["cfg"]))
    spec = json.loads(str(z["spec"]))
    sd = fill_by_name(spec, seed=5, gain=1.0)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    x = torch.from_numpy(z["x"])
    taps, info = {}, {}
    with torch.inference_mode():
        y, _, _ = model_ref.forward(cfg, sd, x, fused=False, taps=taps, moe_info=info)
    assert tuple(y.shape) == tuple(z["y_shape"])
    # discrete decisions first: routed experts of the gated blocks and top-k experts per token of every MoT block
    for k in [f for f in z.files if f.startswith("route::")]:
        assert np.array_equal(info[k[len("route::"):]]["indices"].numpy().astype(np.int16), z[k]), k
    # bit-identical at generation time (asserted by the generator); only the CPU thread count may differ here
    n = len(cfg["backbone"]) + len(cfg["head"])
    for i in range(n - 1):
        got = taps[i].reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        np.testing.assert_allclose(got, z[f"layer{i}_val"], rtol=1e-4, atol=1e-4, err_msg=f"layer {i} ({(cfg['backbone'] + cfg['head'])[i][2]})")
    got = y.reshape(-1)[torch.from_numpy(z["y_idx"].astype(np.int64))].numpy()
    np.testing.assert_allclose(got, z["y_val"], rtol=1e-4, atol=1e-3)
    kinds = {(cfg["backbone"] + cfg["head"])[i][2] for i in range(n)}
    assert {"VisualEnhancedAdaptiveGateMoE", "C2fMoA", "C2fMoT", "A2C2f", "C3k2", "Detect"} <= kinds


def test_shared_expert_pool_model_oracle_reproduces_reference(golden_dir):
    """`v0_8/det/yolo-master-moe-mot-shared-n.yaml` (SharedExpertMoE, moe/shared_expert_moe.py): the P3 and P4 blocks alias ONE expert group;
    the state_dict lists it under both prefixes with (here) different seeded values, and the reference computes BOTH blocks with the
    last member's entries (load_state_dict order).  The oracle restates that rule on the dict; the fixture comes from the real reference
    (tests/golden/make_golden_cfg5.py v08s, bit-exact there)."""
    from oracle import model_ref

    z = np.load(golden_dir / "fwd_v08s.npz")
    cfg = json.loads(str(z["cfg"]))
    sd = fill_by_name(json.loads(str(z["spec"])), seed=5, gain=1.0)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    rows = cfg["backbone"] + cfg["head"]
    members = [i for i, r in enumerate(rows) if r[2] == "SharedExpertMoE"]
    assert len(members) == 2 and rows[members[0]][3][13] == rows[members[1]][3][13]
    a, b = (f"model.{i}.fused_experts.fused.fused_conv.weight" for i in members)
    assert not torch.equal(sd[a], sd[b]), "the fixture must distinguish the two prefixes of the shared tensors"
    x = torch.from_numpy(z["x"])
    taps, info = {}, {}
    with torch.inference_mode():
        y, _, _ = model_ref.forward(cfg, sd, x, fused=False, taps=taps, moe_info=info)
    for k in [f for f in z.files if f.startswith("route::")]:
        assert np.array_equal(info[k[len("route::"):]]["indices"].numpy().astype(np.int16), z[k]), k
    n = len(rows)
    for i in range(n - 1):
        got = taps[i].reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        np.testing.assert_allclose(got, z[f"layer{i}_val"], rtol=1e-4, atol=1e-4, err_msg=f"layer {i} ({rows[i][2]})")
    got = y.reshape(-1)[torch.from_numpy(z["y_idx"].astype(np.int64))].numpy()
    np.testing.assert_allclose(got, z["y_val"], rtol=1e-4, atol=1e-3)
    # ... and the rule is what makes it so: with the FIRST member's entries for both blocks the P3 block's output moves
    sd2 = dict(sd)
    for k in [k for k in sd if k.startswith(f"model.{members[0]}.fused_experts.")]:
        sd2[f"model.{members[1]}.fused_experts." + k[len(f"model.{members[0]}.fused_experts."):]] = sd[k]
    t2 = {}
    with torch.inference_mode():
        model_ref.forward(cfg, sd2, x, fused=False, taps=t2)
    assert not torch.allclose(t2[members[0]], taps[members[0]], rtol=1e-3, atol=1e-3)


def test_ultra_optimized_moe_oracle_reproduces_reference(golden_dir):
    """UltraOptimizedMoE (moe/modules.py:121-232): the five module fixtures of tests/golden/make_golden_uomoe.py (bit-exact there against the real
    reference, incl. a route dropped by the 0.01 inference threshold) and the whole `v0_1/det/yolo-master-n-uomoe.yaml` detector."""
    from oracle import model_ref, uomoe_ref

    for name in ("base", "e16", "widen", "small", "thr"):
        z = np.load(golden_dir / f"uomoe_{name}.npz")
        sd = {f"m.{k}": torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}
        info = {}
        with torch.inference_mode():
            y = uomoe_ref.ultra_optimized_moe(sd, "m", torch.from_numpy(z["x"]), top_k=int(z["args"][3]), info=info)
        assert np.array_equal(info["m"]["indices"].numpy().astype(np.int32), z["indices"]), name
        np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-4, atol=1e-4, err_msg=name)
        if name == "thr":
            assert int((info["m"]["weights"] <= 0.01).sum()) > 0
    z = np.load(golden_dir / "fwd_uomoe.npz")
    cfg = json.loads(str(z["cfg"]))
    sd = fill_by_name(json.loads(str(z["spec"])), seed=5, gain=1.0)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    taps, info = {}, {}
    with torch.inference_mode():
        y, _, _ = model_ref.forward(cfg, sd, torch.from_numpy(z["x"]), fused=False, taps=taps, moe_info=info)
    rows = cfg["backbone"] + cfg["head"]
    assert sum(r[2] == "UltraOptimizedMoE" for r in rows) == 3
    for k in [f for f in z.files if f.startswith("route::")]:
        assert np.array_equal(info[k[len("route::"):]]["indices"].numpy().astype(np.int16), z[k]), k
    for i in range(len(rows) - 1):
        got = taps[i].reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        np.testing.assert_allclose(got, z[f"layer{i}_val"], rtol=1e-4, atol=1e-4, err_msg=f"layer {i} ({rows[i][2]})")
