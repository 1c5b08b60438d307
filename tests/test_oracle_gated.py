"""Gated-MoE oracle (`VisualEnhancedAdaptiveGateMoE`) against golden vectors produced by the REAL
reference module (tests/golden/make_golden_gated.py).  CPU only; the reference is not needed at run time."""
import numpy as np
import pytest
import torch

from oracle import gated_ref

CASES = {"base": {}, "small": {}, "keep1": {}, "e6k3": dict(num_experts=6, top_k=3), "mid": {},
         "e16": dict(num_experts=16, top_k=2)}   # more experts than the fused threshold: shared-inverted expert backend


def _load(golden_dir, name):
    z = np.load(golden_dir / f"gated_{name}.npz")
    sd = {f"m.{k}": torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}
    return sd, z


@pytest.mark.parametrize("name", list(CASES))
def test_gated_oracle_reproduces_reference(name, golden_dir):
    sd, z = _load(golden_dir, name)
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    info = {}
    with torch.inference_mode():
        out = gated_ref.visual_enhanced_moe(sd, "m", x, info=info, **CASES[name])
    # bit-identical at generation time (asserted by make_golden_gated.py); only the CPU thread count may differ here
    assert float((out - y).abs().max()) <= 1e-5 * float(y.abs().max()), f"max |dy| = {(out - y).abs().max().item():.3e}"
    r = info["m"]
    assert np.array_equal(r["indices"].numpy(), z["indices"]), "routed experts differ from the reference"
    assert float(np.abs(r["weights"].numpy() - z["weights"]).max()) <= 1e-6
    assert abs(float(r["complexity"]) - float(z["complexity"])) <= 1e-6
    w = r["weights"].view(x.shape[0], -1)
    assert torch.allclose(w.sum(1), torch.ones(x.shape[0]), atol=1e-5)
    keep = int(torch.round(r["complexity"].clamp(0.3, 1.5) * w.shape[1]).clamp(1, w.shape[1]))
    assert bool(((w > 0).sum(1) <= keep).all()), "more experts active than the complexity gate allows"


def test_gated_structural_properties(golden_dir):
    sd, z = _load(golden_dir, "base")
    x = torch.from_numpy(z["x"])
    # complexity gate: c = 0.3 keeps one of two, c = 1.0 keeps both, c = 1.5 cannot exceed top_k
    w = torch.tensor([[0.7, 0.3]]).view(1, 2, 1, 1)
    assert gated_ref.complexity_gate(w, torch.tensor(0.3)).view(-1).tolist() == [1.0, 0.0]
    assert torch.allclose(gated_ref.complexity_gate(w, torch.tensor(1.0)).view(-1), torch.tensor([0.7, 0.3]))
    assert torch.allclose(gated_ref.complexity_gate(w, torch.tensor(1.5)).view(-1), torch.tensor([0.7, 0.3]))
    # the router is a per-image decision: permuting the batch permutes weights and indices
    perm = torch.tensor([2, 0, 3, 1])
    xd = x[:, 32:]
    w0, i0, _ = gated_ref.dual_stream_router(sd, "m.routing", xd, 2, 1.2)
    w1, i1, _ = gated_ref.dual_stream_router(sd, "m.routing", xd[perm], 2, 1.2)
    assert torch.equal(i0[perm], i1) and torch.allclose(w0[perm], w1, atol=1e-6)
    # detail gate is the identity when its scale is zero
    sd0 = dict(sd)
    sd0["m.detail_gate.detail_scale"] = torch.tensor(0.0)
    assert torch.equal(gated_ref.detail_gate(sd0, "m.detail_gate", xd), xd)


@pytest.mark.parametrize("name", ["opt_base", "opt_e16", "fus_base", "fus_small", "fus_e16", "fus_keep1", "mh_base", "mh_e16", "mh_h3", "div_base", "div_e16", "div_k3", "div_keep1"])
def test_v12_v15_restatement_matches_reference(name, golden_dir):
    """oracle optimal_hybrid_moe (OptimalHybridGateMoE v0_12 / GatedFusionMoE v0_15) against the real reference's vectors."""
    import numpy as np

    z = np.load(golden_dir / f"gated2_{name}.npz")
    sd = {f"m.{k}": torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}
    kw = eval(str(z["kw"]), {"__builtins__": {}}, {"dict": dict})
    info = {}
    with torch.inference_mode():
        y = gated_ref.optimal_hybrid_moe(sd, "m", torch.from_numpy(z["x"]), info=info, cross_gate=str(z["cls"]) == "GatedFusionMoE",
                                         **{k: v for k, v in kw.items() if k in ("num_experts", "top_k", "split_ratio")})
    assert np.array_equal(info["m"]["indices"].numpy(), z["indices"])
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["agm", "agm_hooks", "agm_keep1", "fused", "hyb", "hyb_e16", "hyb2", "lowrank", "refined", "detail", "ctxref"])
def test_chain_restatement_matches_reference(name, golden_dir):
    """oracle adaptive_gate_chain (AdaptiveGateMoE v0_4 ... ContextRefined, HybridAdaptiveGateMoEv2 v0_11) against the real
    reference's vectors (tests/golden/make_golden_gated.py chain)."""
    import numpy as np

    z = np.load(golden_dir / f"gated3_{name}.npz")
    sd = {f"m.{k}": torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}
    okw = eval(str(z["okw"]), {"__builtins__": {}}, {"dict": dict})
    info = {}
    with torch.inference_mode():
        y = gated_ref.adaptive_gate_chain(sd, "m", torch.from_numpy(z["x"]), info=info, **okw)
    assert np.array_equal(info["m"]["indices"].numpy(), z["indices"])
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-5, atol=1e-5)
