"""Golden vectors for the gated-MoE oracle (oracle/gated_ref.py), produced by the REAL reference module
`VisualEnhancedAdaptiveGateMoE` on CPU.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_gated.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
from oracle import gated_ref, refboot  # noqa: E402

refboot.boot()
from make_golden_moa import seeded_fill  # noqa: E402

from ultralytics.nn.modules.moe import gated as ref_gated  # noqa: E402
from ultralytics.nn.modules.moe.gated import GatedFusionMoE, OptimalHybridGateMoE, VisualEnhancedAdaptiveGateMoE  # noqa: E402


def case_chain(name, cls_name, C, x, seed, tweak=None, **kw):
    """Earlier members of the AdaptiveGateMoE chain (v0_4 ... v0_9, v0_11) -> gated3_<name>.npz (python make_golden_gated.py chain)."""
    cls = getattr(ref_gated, cls_name)
    m = cls(C, C, **kw)
    sd = seeded_fill(m, seed)
    if tweak:
        tweak(sd)
        m.load_state_dict(sd)
    m.eval()
    plain = cls_name in ("AdaptiveGateMoE", "FusedAdaptiveGateMoE")      # their forward: no channel shuffle, complexity before the hooks
    okw = dict(num_experts=kw.get("num_experts", 4), top_k=kw.get("top_k", 2), split_ratio=kw.get("split_ratio", 0.5),
               temperature=float(m.routing.temperature), shuffle_groups=1 if plain else 2, complexity_after_hooks=not plain,
               hooks=list(kw["router_hooks"]) if kw.get("router_hooks") else None)
    info = {}
    with torch.inference_mode():
        y = m(x)
        oy = gated_ref.adaptive_gate_chain({f"m.{k}": v for k, v in sd.items()}, "m", x, info=info, **okw)
    exact = torch.equal(y, oy)
    r = info["m"]
    print(f"[gated3_{name}] {cls_name} x {tuple(x.shape)} backend {getattr(m, 'expert_backend', 'shared_inverted')} hooks {m.router_hook_names}; "
          f"oracle bit-exact vs reference: {exact}; max|dy| {(y - oy).abs().max().item():.3e}; |y| max {y.abs().max().item():.3f}; "
          f"complexity {float(r['complexity']):.3f}; experts {r['indices'].view(x.shape[0], -1).tolist()}")
    assert exact
    rec = {"x": x.numpy(), "y": y.numpy(), "keys": np.array(list(sd.keys())), "weights": r["weights"].numpy(),
           "indices": r["indices"].numpy(), "complexity": np.float32(r["complexity"]), "cls": np.array(cls_name),
           "kw": np.array(repr(kw)), "okw": np.array(repr(okw))}
    rec.update({f"sd::{k}": v.numpy() for k, v in sd.items()})
    np.savez_compressed(HERE / f"gated3_{name}.npz", **rec)


def case_v2(name, cls, C, x, seed, tweak=None, **kw):
    """OptimalHybridGateMoE (v0_12) / GatedFusionMoE (v0_15) -> gated2_<name>.npz."""
    m = cls(C, C, **kw)
    sd = seeded_fill(m, seed)
    if tweak:
        tweak(sd)
        m.load_state_dict(sd)
    m.eval()
    info = {}
    with torch.inference_mode():
        y = m(x)
        oy = gated_ref.optimal_hybrid_moe({f"m.{k}": v for k, v in sd.items()}, "m", x, info=info, cross_gate=cls is GatedFusionMoE,
                                          **{k: v for k, v in kw.items() if k in ("num_experts", "top_k", "split_ratio")})
    exact = torch.equal(y, oy)
    r = info["m"]
    print(f"[gated2_{name}] {cls.__name__} x {tuple(x.shape)}; oracle bit-exact vs reference: {exact}; max|dy| {(y - oy).abs().max().item():.3e}; "
          f"|y| max {y.abs().max().item():.3f}; complexity {float(r['complexity']):.3f}; experts {r['indices'].view(x.shape[0], -1).tolist()}")
    assert exact
    rec = {"x": x.numpy(), "y": y.numpy(), "keys": np.array(list(sd.keys())), "weights": r["weights"].numpy(),
           "indices": r["indices"].numpy(), "complexity": np.float32(r["complexity"]), "cls": np.array(cls.__name__),
           "kw": np.array(repr(kw))}
    rec.update({f"sd::{k}": v.numpy() for k, v in sd.items()})
    np.savez_compressed(HERE / f"gated2_{name}.npz", **rec)


def case(name, C, x, seed, tweak=None, **kw):
    m = VisualEnhancedAdaptiveGateMoE(C, C, **kw)
    sd = seeded_fill(m, seed)
    if tweak:
        tweak(sd)
        m.load_state_dict(sd)
    m.eval()
    info = {}
    with torch.inference_mode():
        y = m(x)
        oy = gated_ref.visual_enhanced_moe({f"m.{k}": v for k, v in sd.items()}, "m", x, info=info,
                                           **{k: v for k, v in kw.items() if k in ("num_experts", "top_k")})
    exact = torch.equal(y, oy)
    r = info["m"]
    print(f"[gated_{name}] x {tuple(x.shape)}; oracle bit-exact vs reference: {exact}; max|dy| {(y - oy).abs().max().item():.3e}; "
          f"|y| max {y.abs().max().item():.3f}; complexity {float(r['complexity']):.3f}; experts {r['indices'].view(x.shape[0], -1).tolist()} "
          f"weights {[[round(float(v), 3) for v in row] for row in r['weights'].view(x.shape[0], -1)]}")
    assert exact
    rec = {"x": x.numpy(), "y": y.numpy(), "keys": np.array(list(sd.keys())), "weights": r["weights"].numpy(),
           "indices": r["indices"].numpy(), "complexity": np.float32(r["complexity"])}
    rec.update({f"sd::{k}": v.numpy() for k, v in sd.items()})
    np.savez_compressed(HERE / f"gated_{name}.npz", **rec)


if __name__ == "__main__":
    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(777)
    V2_ONLY = len(sys.argv) > 1 and sys.argv[1] == "v2"

    def varied(B, C, H, W):   # images with different channel statistics so that the router separates them
        x = torch.randn(B, C, H, W, generator=g)
        return x * (0.5 + torch.rand(B, C, 1, 1, generator=g) * 2.0) + torch.randn(B, C, 1, 1, generator=g) * 1.5

    def high_complexity(sd):   # complexity -> 1.0: both routed experts kept with their softmax weights
        sd["complexity_estimator.1.bias"] = torch.tensor([20.0])
        if "routing.global_fc.weight" in sd:
            sd["routing.global_fc.weight"] = sd["routing.global_fc.weight"] * 4.0

    def low_complexity(sd):    # complexity clamps to 0.3 -> round(0.6) = 1 expert kept of the top-2
        sd["complexity_estimator.1.bias"] = torch.tensor([-20.0])

    def live_gates(sd):        # the reference initialises gate_scale = 0 and refine_scale = 0.1: make both paths matter
        high_complexity(sd)
        if "cross_gate.gate_scale" in sd:
            sd["cross_gate.gate_scale"] = torch.tensor(0.8)
        sd["refine_scale"] = torch.tensor(0.6)
        sd["routing.expert_prior"] = torch.randn(sd["routing.expert_prior"].shape, generator=torch.Generator().manual_seed(3)) * 0.5

    if len(sys.argv) > 1 and sys.argv[1] == "chain":
        def live(sd):          # both routed experts kept; hook scales away from their near-identity initial values
            high_complexity(sd)
            for k in ("refine_scale", "detail_gate.detail_scale", "context_mixer.context_scale"):
                if k in sd:
                    sd[k] = torch.tensor(0.7)
            if "routing.expert_prior" in sd:
                sd["routing.expert_prior"] = torch.randn(sd["routing.expert_prior"].shape, generator=torch.Generator().manual_seed(5)) * 0.5

        case_chain("agm", "AdaptiveGateMoE", 64, varied(3, 64, 12, 16), 21, tweak=live)
        case_chain("agm_hooks", "AdaptiveGateMoE", 64, varied(3, 64, 10, 10), 22, tweak=live, router_hooks=["refine", "detail"])
        case_chain("agm_keep1", "AdaptiveGateMoE", 64, varied(2, 64, 6, 5), 23, tweak=low_complexity, num_experts=8)
        case_chain("fused", "FusedAdaptiveGateMoE", 64, varied(3, 64, 12, 16), 24, tweak=live)
        case_chain("hyb", "HybridAdaptiveGateMoE", 64, varied(3, 64, 12, 16), 25, tweak=live)
        case_chain("hyb_e16", "HybridAdaptiveGateMoE", 64, varied(3, 64, 8, 8), 26, tweak=live, num_experts=16, top_k=2)
        case_chain("hyb2", "HybridAdaptiveGateMoEv2", 64, varied(3, 64, 12, 16), 27, tweak=live, split_ratio=0.375)
        case_chain("lowrank", "LowRankHybridAdaptiveGateMoE", 64, varied(3, 64, 12, 16), 28, tweak=live)
        case_chain("refined", "RefinedLowRankHybridAdaptiveGateMoE", 64, varied(3, 64, 12, 16), 29, tweak=live)
        case_chain("detail", "DetailAwareLowRankHybridAdaptiveGateMoE", 64, varied(3, 64, 12, 16), 30, tweak=live)
        case_chain("ctxref", "ContextRefinedLowRankHybridAdaptiveGateMoE", 64, varied(3, 64, 12, 16), 31, tweak=live)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "v13":    # MultiHeadRouterMoE (v0_13) -> gated2_mh_*.npz
        def live13(sd):
            live_gates(sd)
            g13 = torch.Generator().manual_seed(13)
            sd["routing.global_proj.weight"] = sd["routing.global_proj.weight"] * 4.0
            for k in [k for k in sd if k.startswith("routing.heads.")]:
                sd[k] = sd[k] * 4.0
            sd["routing.head_alpha"] = torch.randn(sd["routing.head_alpha"].shape, generator=g13)
            sd["routing.global_weight"] = torch.tensor(0.4)

        case_v2("mh_base", ref_gated.MultiHeadRouterMoE, 128, varied(3, 128, 12, 16), 41, tweak=live13)
        case_v2("mh_e16", ref_gated.MultiHeadRouterMoE, 128, varied(3, 128, 8, 8), 42, tweak=live13, num_experts=16, top_k=2, split_ratio=0.375)
        case_v2("mh_h3", ref_gated.MultiHeadRouterMoE, 128, varied(2, 128, 6, 7), 43, tweak=live13, num_experts=6, top_k=3, num_heads=3)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "v14":    # DiversifiedExpertMoE (v0_14) -> gated2_div_*.npz
        case_v2("div_base", ref_gated.DiversifiedExpertMoE, 128, varied(3, 128, 12, 16), 51, tweak=live_gates)
        case_v2("div_e16", ref_gated.DiversifiedExpertMoE, 128, varied(3, 128, 20, 18), 52, tweak=live_gates, num_experts=16, top_k=2, split_ratio=0.375)
        case_v2("div_k3", ref_gated.DiversifiedExpertMoE, 128, varied(2, 128, 6, 7), 53, tweak=live_gates, num_experts=6, top_k=3)
        case_v2("div_keep1", ref_gated.DiversifiedExpertMoE, 128, varied(2, 128, 9, 9), 54, tweak=lambda sd: (live_gates(sd), low_complexity(sd)))
        sys.exit(0)
    case_v2("opt_base", OptimalHybridGateMoE, 128, varied(3, 128, 12, 16), 11, tweak=live_gates)
    case_v2("opt_e16", OptimalHybridGateMoE, 128, varied(3, 128, 8, 8), 12, tweak=live_gates, num_experts=16, top_k=2, split_ratio=0.375)
    case_v2("fus_base", GatedFusionMoE, 128, varied(4, 128, 12, 16), 13, tweak=live_gates)
    case_v2("fus_small", GatedFusionMoE, 128, varied(2, 128, 4, 3), 14, tweak=live_gates, num_experts=8, top_k=2)
    case_v2("fus_e16", GatedFusionMoE, 128, varied(3, 128, 8, 10), 15, tweak=live_gates, num_experts=16, top_k=2, split_ratio=0.375)
    case_v2("fus_keep1", GatedFusionMoE, 128, varied(2, 128, 9, 9), 16, tweak=lambda sd: (live_gates(sd), low_complexity(sd)))
    if V2_ONLY:
        sys.exit(0)
    case("base", 64, varied(4, 64, 16, 20), 1, tweak=high_complexity)
    case("small", 64, varied(3, 64, 4, 3), 2, tweak=high_complexity)                   # map not larger than the router pool
    case("keep1", 64, varied(2, 64, 12, 12), 3, tweak=low_complexity)
    case("e6k3", 96, varied(3, 96, 10, 14), 4, tweak=high_complexity, num_experts=6, top_k=3)
    case("e16", 64, varied(4, 64, 8, 10), 6, tweak=high_complexity, num_experts=16, top_k=2)   # shared-inverted backend (E > 8)
    case("mid", 64, varied(2, 64, 9, 9), 5)                                            # untouched estimator (complexity ~0.5)
