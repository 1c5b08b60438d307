"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the eval-mode forward of `VisualEnhancedAdaptiveGateMoE` (the gated-MoE
family behind the v0_10 master YAMLs, i.e. the backbone of the shipped MoA/MoT configurations).  The class is the end
of an inheritance chain (AdaptiveGateMoE -> Hybrid -> LowRankHybrid -> Refined -> ContextRefined -> VisualEnhanced,
`nn/modules/moe/gated.py:268-1764`) whose forward is `run_visual_hybrid_moe_forward`
(`nn/modules/moe/_gated_visual.py:32-96`) with the three router hooks detail / context / refine
(`nn/modules/moe/hooks.py`).  Every function cites the lines it follows; parity is pinned by
`tests/golden/make_golden_gated.py` (REAL reference on CPU) and checked by `tests/test_oracle_gated.py`.

State-dict layout: the reference's own parameter names under a module prefix `p`.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .moa_ref import safe_groups
from .model_ref import BN_EPS


def _gn(sd, p, x, desired):
    w = sd[f"{p}.weight"]
    return F.group_norm(x, safe_groups(w.shape[0], desired), w, sd[f"{p}.bias"], 1e-5)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"], False, 0.0,
                        BN_EPS)


def _dw3(x, w):
    return F.conv2d(x, w, None, 1, 1, 1, x.shape[1])


def dual_stream_router(sd, p, x, top_k, temperature, pool_scale=4):
    """DualStreamGateRouter.forward (gated.py:124-166), fp32: global stream = Linear over [mean, std] channel
    statistics; local stream = 4x4 average pool (when the map is larger than the pool) -> DW3x3 -> GN(<=8) -> SiLU
    -> 1x1 -> GN(<=4) -> SiLU -> 1x1(+bias) -> spatial mean; sigmoid(alpha) blend, clamp +-30, softmax(logits / T),
    top-k, renormalise (+1e-6).  Returns (weights [B,k,1,1] in x.dtype, indices [B,k,1,1], probs)."""
    B, C, H, W = x.shape
    xf = x.float()
    mean = xf.mean(dim=[2, 3])
    std = xf.std(dim=[2, 3], unbiased=False) if H * W > 1 else torch.zeros_like(mean)
    g = F.linear(torch.cat([mean, std], dim=1), sd[f"{p}.global_fc.weight"])
    xl = F.avg_pool2d(xf, kernel_size=pool_scale, stride=pool_scale) if (H > pool_scale and W > pool_scale) else xf
    h = F.silu(_gn(sd, f"{p}.local_conv.1", _dw3(xl, sd[f"{p}.local_conv.0.weight"]), 8))
    h = F.silu(_gn(sd, f"{p}.local_conv.4", F.conv2d(h, sd[f"{p}.local_conv.3.weight"]), 4))
    loc = F.conv2d(h, sd[f"{p}.local_conv.6.weight"], sd[f"{p}.local_conv.6.bias"]).mean(dim=[2, 3])
    a = torch.sigmoid(sd[f"{p}.alpha"])
    logits = (a * g + (1 - a) * loc).clamp(-30.0, 30.0)
    probs = F.softmax(logits / max(float(temperature), 1e-3), dim=1)
    tw, ti = torch.topk(probs, top_k, dim=1)
    tw = tw / (tw.sum(dim=1, keepdim=True) + 1e-6)
    return tw.to(x.dtype).view(B, top_k, 1, 1), ti.view(B, top_k, 1, 1), probs


def complexity_gate(weights, complexity):
    """AdaptiveGateMoE._apply_complexity_gate (gated.py:462-492): a batch-level complexity score in [0.3, 1.5]
    keeps round(c * top_k) (clamped to [1, top_k]) of the ranked experts; the kept weights are renormalised."""
    k = weights.shape[1]
    if k <= 1:
        return weights
    c = torch.nan_to_num(complexity, nan=1.0, posinf=1.0, neginf=1.0).clamp(0.3, 1.5)
    keep = torch.round(c * k).clamp(1, k)
    rank = torch.arange(1, k + 1, dtype=keep.dtype)
    w = weights * (rank.view(1, k, 1, 1) <= keep).to(weights.dtype)
    return w / w.sum(dim=1, keepdim=True).clamp_min(1e-6)


def fused_experts(sd, p, x, weights, indices, num_experts, num_groups=8):
    """LowRankFusedExpertGroup / FusedExpertGroup.forward (gated.py:1058-1090, 1144-1146): shared 1x1 bottleneck
    (GN, SiLU); one grouped 3x3 conv produces all experts' outputs [B, E, OC, H, W]; the routed top-k are gathered,
    group-normalised without affine, scaled by the selected expert's affine row, SiLU, weighted sum."""
    h = F.silu(_gn(sd, f"{p}.bottleneck.1", F.conv2d(x, sd[f"{p}.bottleneck.0.weight"]), num_groups))
    wf = sd[f"{p}.fused.fused_conv.weight"]
    groups = h.shape[1] // wf.shape[1]
    B, _, H, W = h.shape
    E, OC = num_experts, wf.shape[0] // num_experts
    k = weights.shape[1]
    f = F.conv2d(h, wf, None, 1, 1, 1, groups).view(B, E, OC, H, W)
    idx = indices.view(B, k)
    sel = torch.gather(f, 1, idx.view(B, k, 1, 1, 1).expand(B, k, OC, H, W))
    ws = sd[f"{p}.fused.expert_norm_weight"][idx].to(f.dtype)
    bs = sd[f"{p}.fused.expert_norm_bias"][idx].to(f.dtype)
    n = F.group_norm(sel.reshape(B * k, OC, H, W), safe_groups(OC, num_groups), None, None, 1e-5).view(B, k, OC, H, W)
    n = F.silu(n * ws.view(B, k, OC, 1, 1) + bs.view(B, k, OC, 1, 1))
    return (n * weights.view(B, k, 1, 1, 1)).sum(dim=1)


def shared_inverted_experts(sd, p, x, weights, indices):
    """SharedInvertedExpertGroup.forward, eager sparse path (moe/experts.py:235-269): shared 1x1 expand -> GN(<=8) ->
    SiLU -> DW3x3 -> GN -> SiLU once; every ACTIVE expert (ascending index) projects (1x1 -> GN) only the images
    that routed to it with a positive weight and is accumulated with index_add_ in that order."""
    B, _, H, W = x.shape
    k = weights.shape[1]
    h = F.silu(_gn(sd, f"{p}.shared_feature.1", F.conv2d(x, sd[f"{p}.shared_feature.0.weight"]), 8))
    wd = sd[f"{p}.shared_feature.3.weight"]
    h = F.silu(_gn(sd, f"{p}.shared_feature.4", F.conv2d(h, wd, None, 1, wd.shape[-1] // 2, 1, h.shape[1]), 8))
    idx = indices.reshape(B, -1)[:, :k].to(torch.long)
    w = weights.reshape(B, -1)[:, :k]
    valid = w > 0.0
    out = x.new_zeros(B, sd[f"{p}.expert_projections.0.0.weight"].shape[0], H, W)
    for e in torch.unique(idx[valid]).to(torch.long).tolist():
        bi, ki = torch.where((idx == e) & valid)
        eo = _gn(sd, f"{p}.expert_projections.{e}.1", F.conv2d(h[bi], sd[f"{p}.expert_projections.{e}.0.weight"]), 8)
        out.index_add_(0, bi, (eo * w[bi, ki].view(-1, 1, 1, 1).to(eo.dtype)).to(out.dtype))
    return out


def detail_gate(sd, p, x, num_groups=8):
    """VisualDetailGate.forward (gated.py:1171-1175): high-pass (x - 3x3 mean) -> DW3x3, GN, SiLU, 1x1, SiLU,
    1x1(+bias), sigmoid; x * (1 + tanh(scale) * gate)."""
    d = x - F.avg_pool2d(x, kernel_size=3, stride=1, padding=1)
    h = F.silu(_gn(sd, f"{p}.detail_filter.1", _dw3(d, sd[f"{p}.detail_filter.0.weight"]), num_groups))
    h = F.silu(F.conv2d(h, sd[f"{p}.detail_filter.3.weight"]))
    g = torch.sigmoid(F.conv2d(h, sd[f"{p}.detail_filter.5.weight"], sd[f"{p}.detail_filter.5.bias"]))
    return x * (1 + torch.tanh(sd[f"{p}.detail_scale"]) * g)


def context_mixer(sd, p, x, num_groups=8, pool_scales=(2, 4)):
    """PyramidContextMixer.forward (gated.py:1209-1218): mean of a local DW3x3 branch and nearest-upsampled 1x1
    projections of adaptive average pools at 1/2 and 1/4; gated residual."""
    B, C, H, W = x.shape
    ctx = [F.silu(_gn(sd, f"{p}.local_context.1", _dw3(x, sd[f"{p}.local_context.0.weight"]), num_groups))]
    for i, s in enumerate(pool_scales):
        h, w = max(1, H // s), max(1, W // s)
        pooled = x if (H, W) == (h, w) else F.adaptive_avg_pool2d(x, (h, w))
        pr = F.silu(_gn(sd, f"{p}.pool_projections.{i}.1", F.conv2d(pooled, sd[f"{p}.pool_projections.{i}.0.weight"]), num_groups))
        ctx.append(F.interpolate(pr, size=(H, W), mode="nearest"))
    c = torch.stack(ctx, dim=0).mean(dim=0)
    gate = torch.sigmoid(F.conv2d(c, sd[f"{p}.context_gate.0.weight"], sd[f"{p}.context_gate.0.bias"]))
    return x + torch.tanh(sd[f"{p}.context_scale"]) * c * gate


def refine(sd, p, x, num_groups=8):
    """FeatureRefinementHook (hooks.py:60-68): x + tanh(scale) * (DW3x3, GN, SiLU)(x) * sigmoid(1x1(SiLU(1x1(GAP x))))."""
    r = F.silu(_gn(sd, f"{p}.feature_refiner.1", _dw3(x, sd[f"{p}.feature_refiner.0.weight"]), num_groups))
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.silu(F.conv2d(g, sd[f"{p}.feature_gate.1.weight"]))
    g = torch.sigmoid(F.conv2d(g, sd[f"{p}.feature_gate.3.weight"], sd[f"{p}.feature_gate.3.bias"]))
    return x + sd[f"{p}.refine_scale"].tanh() * r * g


def adaptive_gate_chain(sd, p, x, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, temperature=1.2, shuffle_groups=2,
                        hooks=None, info=None, complexity_after_hooks=True):
    """Eval forward of every member of the AdaptiveGateMoE inheritance chain (gated.py:268-1764): AdaptiveGateMoE v0_4 (:530-580;
    shared-inverted experts, no shuffle), FusedAdaptiveGateMoE v0_5 (:1232-1274; FusedExpertGroup), HybridAdaptiveGateMoE v0_6
    (:1340-1386; fused for E <= 8 else shared-inverted, channel shuffle), HybridAdaptiveGateMoEv2 v0_11 (:1389-1452; router V2),
    LowRankHybridAdaptiveGateMoE v0_7 (:1455-1508; bottlenecked fused experts), Refined v0_8 / DetailAware v0_9 / ContextRefined /
    VisualEnhanced v0_10 (:1511-1764, `run_visual_hybrid_moe_forward`, _gated_visual.py:32-75, with the router hooks
    detail -> pre_route, context / refine -> post_fusion in declaration order).

    The members differ only in which sub-modules exist, so the state dict decides: `routing.stat_norm` -> router V2,
    `fused_experts.shared_feature` / `.bottleneck` / `.fused_conv` -> expert backend, `detail_gate` / `context_mixer` /
    `feature_refiner` -> hooks (`hooks` overrides the order for AdaptiveGateMoE(router_hooks=[...])).  `shuffle_groups` is 1 for
    v0_4 / v0_5 (their forward has no _channel_shuffle) and `temperature` is the router's eval temperature (1.0 for v0_4 / v0_5).
    The complexity score reads the dynamic half after the pre-route hooks in `run_visual_hybrid_moe_forward` (_gated_visual.py:47-53)
    and before them in AdaptiveGateMoE.forward (gated.py:548-552): `complexity_after_hooks`."""
    B, C, H, W = x.shape
    dyn = int(C * split_ratio)
    st = C - dyn
    if hooks is None:
        hooks = [h for h, key in (("detail", f"{p}.detail_gate.detail_scale"), ("context", f"{p}.context_mixer.context_scale"),
                                  ("refine", f"{p}.feature_refiner.0.weight")) if key in sd]
    # SE gate (gated.py:333-341): GAP -> Linear (no bias) -> SiLU -> Linear -> sigmoid
    g = F.adaptive_avg_pool2d(x, 1).flatten(1)
    g = torch.sigmoid(F.linear(F.silu(F.linear(g, sd[f"{p}.se_gate.2.weight"])), sd[f"{p}.se_gate.4.weight"], sd[f"{p}.se_gate.4.bias"]))
    xs = x[:, :st] * g[:, :st].unsqueeze(-1).unsqueeze(-1)
    xd = x[:, st:] * g[:, st:].unsqueeze(-1).unsqueeze(-1)
    # static path (gated.py:344-353): DW3x3 -> BN -> SiLU -> 1x1 -> BN -> SiLU
    s = F.silu(_bn(sd, f"{p}.static_net.1", _dw3(xs, sd[f"{p}.static_net.0.weight"])))
    s = F.silu(_bn(sd, f"{p}.static_net.4", F.conv2d(s, sd[f"{p}.static_net.3.weight"])))
    def complexity(t):   # gated.py:373-377, 455-460
        c = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(t, 1), sd[f"{p}.complexity_estimator.1.weight"],
                                   sd[f"{p}.complexity_estimator.1.bias"])).mean()
        return torch.tensor(1.0) if (torch.isnan(c) or torch.isinf(c)) else c.clamp(0.3, 1.5)

    cplx = None if complexity_after_hooks else complexity(xd)
    for h in hooks:                                                                 # pre_route hooks
        if h == "detail":
            xd = detail_gate(sd, f"{p}.detail_gate", xd, num_groups)
    if cplx is None:
        cplx = complexity(xd)
    router = dual_stream_router_v2 if f"{p}.routing.stat_norm.weight" in sd else dual_stream_router
    w, idx, probs = router(sd, f"{p}.routing", xd, top_k, temperature)
    w = complexity_gate(w, cplx)
    if info is not None:
        info[p] = {"weights": w, "indices": idx, "probs": probs, "complexity": cplx}
    if f"{p}.fused_experts.shared_feature.0.weight" in sd:   # AdaptiveGateMoE, or more experts than fused_expert_threshold (:1318-1331)
        d = shared_inverted_experts(sd, f"{p}.fused_experts", xd, w, idx)
    elif f"{p}.fused_experts.bottleneck.0.weight" in sd:
        d = fused_experts(sd, f"{p}.fused_experts", xd, w, idx, num_experts, num_groups)
    else:
        d = plain_fused_experts(sd, f"{p}.fused_experts", xd, w, idx, num_experts, num_groups)
    cat = torch.cat([s, d], dim=1)
    oc = cat.shape[1]
    sg = shuffle_groups if oc % shuffle_groups == 0 else 1
    if sg > 1:                                                                      # _channel_shuffle (gated.py:1333-1338)
        cat = cat.view(B, sg, oc // sg, H, W).transpose(1, 2).reshape(B, oc, H, W)
    for h in hooks:                                                                 # post_fusion hooks, declaration order
        if h == "context":
            cat = context_mixer(sd, f"{p}.context_mixer", cat, num_groups)
        elif h == "refine":
            cat = refine(sd, p, cat, num_groups)
    return _gn(sd, f"{p}.bn", F.conv2d(cat, sd[f"{p}.proj.weight"]), num_groups) + x


def visual_enhanced_moe(sd, p, x, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, temperature=1.2,
                        shuffle_groups=2, info=None):
    """run_visual_hybrid_moe_forward for VisualEnhancedAdaptiveGateMoE, eval (_gated_visual.py:32-75)."""
    return adaptive_gate_chain(sd, p, x, num_experts, top_k, split_ratio, num_groups, temperature, shuffle_groups, info=info)


# ---------------------------------------------------------------------------------- v0_12 / v0_15 members of the family
def dual_stream_router_v2(sd, p, x, top_k, temperature, pool_scale=4):
    """DualStreamGateRouterV2.forward, eval (gated.py:217-262): as dual_stream_router with LayerNorm over the concatenated
    [mean, std] statistics before the global Linear, and the learnable `expert_prior` added to the blended logits."""
    B, C, H, W = x.shape
    xf = x.float()
    mean = xf.mean(dim=[2, 3])
    std = xf.std(dim=[2, 3], unbiased=False) if H * W > 1 else torch.zeros_like(mean)
    stats = F.layer_norm(torch.cat([mean, std], dim=1), (2 * C,), sd[f"{p}.stat_norm.weight"], sd[f"{p}.stat_norm.bias"], 1e-5)
    g = F.linear(stats, sd[f"{p}.global_fc.weight"])
    xl = F.avg_pool2d(xf, kernel_size=pool_scale, stride=pool_scale) if (H > pool_scale and W > pool_scale) else xf
    h = F.silu(_gn(sd, f"{p}.local_conv.1", _dw3(xl, sd[f"{p}.local_conv.0.weight"]), 8))
    h = F.silu(_gn(sd, f"{p}.local_conv.4", F.conv2d(h, sd[f"{p}.local_conv.3.weight"]), 4))
    loc = F.conv2d(h, sd[f"{p}.local_conv.6.weight"], sd[f"{p}.local_conv.6.bias"]).mean(dim=[2, 3])
    a = torch.sigmoid(sd[f"{p}.alpha"])
    logits = a * g + (1 - a) * loc
    logits = (logits + sd[f"{p}.expert_prior"].view(1, -1)).clamp(-30.0, 30.0)
    probs = F.softmax(logits / max(float(temperature), 1e-3), dim=1)
    tw, ti = torch.topk(probs, top_k, dim=1)
    tw = tw / (tw.sum(dim=1, keepdim=True) + 1e-6)
    return tw.to(x.dtype).view(B, top_k, 1, 1), ti.view(B, top_k, 1, 1), probs


def multi_head_router_v3(sd, p, x, top_k, temperature, pool_scale=4):
    """MultiHeadRouterV3.forward, eval (gated.py:2108-2190): the LayerNorm-ed [mean, std] statistics feed one full-width Linear
    (`global_proj`, weight sigmoid(global_weight)) and `num_heads` Linears over consecutive `head_dim` slices of the (zero padded or
    truncated) statistics, mixed by normalised sigmoid(head_alpha); the blend with the local stream, the prior, the clamp, softmax
    and top-k are DualStreamGateRouterV2's."""
    B, C, H, W = x.shape
    xf = x.float()
    mean = xf.mean(dim=[2, 3])
    std = xf.std(dim=[2, 3], unbiased=False) if H * W > 1 else torch.zeros_like(mean)
    stats = F.layer_norm(torch.cat([mean, std], dim=1), (2 * C,), sd[f"{p}.stat_norm.weight"], sd[f"{p}.stat_norm.bias"], 1e-5)
    nh = sd[f"{p}.head_alpha"].numel()
    hd = sd[f"{p}.heads.0.weight"].shape[1]
    hw = torch.sigmoid(sd[f"{p}.head_alpha"])
    hw = hw / (hw.sum() + 1e-6)
    gw = torch.sigmoid(sd[f"{p}.global_weight"])
    sp = F.pad(stats, (0, hd * nh - stats.shape[1])) if stats.shape[1] < hd * nh else stats[:, : hd * nh]
    chunks = sp.view(B, nh, hd)
    hl = gw * F.linear(stats, sd[f"{p}.global_proj.weight"])
    for i in range(nh):
        hl = hl + (1 - gw) * hw[i] * F.linear(chunks[:, i, :], sd[f"{p}.heads.{i}.weight"])
    xl = F.avg_pool2d(xf, kernel_size=pool_scale, stride=pool_scale) if (H > pool_scale and W > pool_scale) else xf
    h = F.silu(_gn(sd, f"{p}.local_conv.1", _dw3(xl, sd[f"{p}.local_conv.0.weight"]), 8))
    h = F.silu(_gn(sd, f"{p}.local_conv.4", F.conv2d(h, sd[f"{p}.local_conv.3.weight"]), 4))
    loc = F.conv2d(h, sd[f"{p}.local_conv.6.weight"], sd[f"{p}.local_conv.6.bias"]).mean(dim=[2, 3])
    a = torch.sigmoid(sd[f"{p}.alpha"])
    logits = a * hl + (1 - a) * loc
    logits = (logits + sd[f"{p}.expert_prior"].view(1, -1)).clamp(-30.0, 30.0)
    probs = F.softmax(logits / max(float(temperature), 1e-3), dim=1)
    tw, ti = torch.topk(probs, top_k, dim=1)
    tw = tw / (tw.sum(dim=1, keepdim=True) + 1e-6)
    return tw.to(x.dtype).view(B, top_k, 1, 1), ti.view(B, top_k, 1, 1), probs


def plain_fused_experts(sd, p, x, weights, indices, num_experts, num_groups=8):
    """FusedExpertGroup.forward (gated.py:1058-1090) on the dynamic half directly (no bottleneck): grouped 3x3 for all experts,
    gather top-k, affine-free GroupNorm, the routed expert's affine row, SiLU, weighted sum."""
    wf = sd[f"{p}.fused_conv.weight"]
    groups = x.shape[1] // wf.shape[1]
    B, _, H, W = x.shape
    E, OC = num_experts, wf.shape[0] // num_experts
    k = weights.shape[1]
    f = F.conv2d(x, wf, None, 1, 1, 1, groups).view(B, E, OC, H, W)
    idx = indices.view(B, k)
    sel = torch.gather(f, 1, idx.view(B, k, 1, 1, 1).expand(B, k, OC, H, W))
    ws = sd[f"{p}.expert_norm_weight"][idx].to(f.dtype)
    bs = sd[f"{p}.expert_norm_bias"][idx].to(f.dtype)
    n = F.group_norm(sel.reshape(B * k, OC, H, W), safe_groups(OC, num_groups), None, None, 1e-5).view(B, k, OC, H, W)
    n = F.silu(n * ws.view(B, k, OC, 1, 1) + bs.view(B, k, OC, 1, 1))
    return (n * weights.view(B, k, 1, 1, 1)).sum(dim=1)


def diversified_experts(sd, p, x, weights, indices, num_groups=8):
    """DiversifiedExpertGroup.forward, eager path (gated.py:2296-2330): shared 1x1 expand -> GN -> SiLU once; every ACTIVE expert
    (ascending index) runs its own dilated depthwise 3x3 (dilation 1 + i // 2) -> GN -> SiLU and its 1x1 projection -> GN on the images
    that routed to it with a positive weight; weighted index_add_ in that order."""
    B, _, H, W = x.shape
    k = weights.shape[1]
    h = F.silu(_gn(sd, f"{p}.shared_expand.1", F.conv2d(x, sd[f"{p}.shared_expand.0.weight"]), num_groups))
    idx = indices.reshape(B, -1)[:, :k].to(torch.long)
    w = weights.reshape(B, -1)[:, :k]
    valid = w > 0.0
    out = x.new_zeros(B, sd[f"{p}.expert_projections.0.0.weight"].shape[0], H, W)
    for e in torch.unique(idx[valid]).to(torch.long).tolist():
        d = 1 + e // 2
        f = F.conv2d(h, sd[f"{p}.dw_layers.{e}.0.weight"], None, 1, d, d, h.shape[1])
        f = F.silu(_gn(sd, f"{p}.dw_layers.{e}.1", f, num_groups))
        bi, ki = torch.where((idx == e) & valid)
        eo = _gn(sd, f"{p}.expert_projections.{e}.1", F.conv2d(f[bi], sd[f"{p}.expert_projections.{e}.0.weight"]), num_groups)
        out.index_add_(0, bi, (eo * w[bi, ki].view(-1, 1, 1, 1).to(eo.dtype)).to(out.dtype))
    return out


def cross_path_gate(sd, p, s, d):
    """CrossPathGate.forward (gated.py:2398-2428): gate = 0.5 + tanh(gate_scale) * 0.5 * sigmoid(MLP(GAP(cat[s, d]))); the first
    Cs / next Cd entries scale the static / dynamic outputs; returns their concatenation."""
    cs, cd = s.shape[1], d.shape[1]
    g = F.adaptive_avg_pool2d(torch.cat([s, d], dim=1), 1).flatten(1)
    raw = F.linear(F.silu(F.linear(g, sd[f"{p}.gate_net.2.weight"])), sd[f"{p}.gate_net.4.weight"], sd[f"{p}.gate_net.4.bias"])
    gate = 0.5 + torch.tanh(sd[f"{p}.gate_scale"]) * 0.5 * torch.sigmoid(raw)
    return torch.cat([s * gate[:, :cs].unsqueeze(-1).unsqueeze(-1), d * gate[:, cs: cs + cd].unsqueeze(-1).unsqueeze(-1)], dim=1)


def optimal_hybrid_moe(sd, p, x, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, temperature=1.2, shuffle_groups=2,
                       cross_gate=False, info=None):
    """OptimalHybridGateMoE.forward (v0_12, gated.py:1953-2008) and, with cross_gate=True, GatedFusionMoE.forward (v0_15,
    gated.py:2630-2693), eval: SE-gated split, static DW+PW path, complexity-gated top-k routing (router V2), fused or
    shared-inverted experts, [cross-path gate,] channel shuffle, residual DW refinement x + tanh(s) * GN(DW3x3(x)) * SE(x),
    1x1 projection, GroupNorm, + x."""
    B, C, H, W = x.shape
    dyn = int(C * split_ratio)
    st = C - dyn
    g = F.adaptive_avg_pool2d(x, 1).flatten(1)
    g = torch.sigmoid(F.linear(F.silu(F.linear(g, sd[f"{p}.se_gate.2.weight"])), sd[f"{p}.se_gate.4.weight"], sd[f"{p}.se_gate.4.bias"]))
    xs = x[:, :st] * g[:, :st].unsqueeze(-1).unsqueeze(-1)
    xd = x[:, st:] * g[:, st:].unsqueeze(-1).unsqueeze(-1)
    s = F.silu(_bn(sd, f"{p}.static_net.1", _dw3(xs, sd[f"{p}.static_net.0.weight"])))
    s = F.silu(_bn(sd, f"{p}.static_net.4", F.conv2d(s, sd[f"{p}.static_net.3.weight"])))
    cplx = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(xd, 1), sd[f"{p}.complexity_estimator.1.weight"],
                                  sd[f"{p}.complexity_estimator.1.bias"])).mean()
    cplx = torch.tensor(1.0) if (torch.isnan(cplx) or torch.isinf(cplx)) else cplx.clamp(0.3, 1.5)
    router = multi_head_router_v3 if f"{p}.routing.heads.0.weight" in sd else dual_stream_router_v2   # MultiHeadRouterMoE (v0_13, :2430-2496)
    w, idx, probs = router(sd, f"{p}.routing", xd, top_k, temperature)
    w = complexity_gate(w, cplx)
    if info is not None:
        info[p] = {"weights": w, "indices": idx, "probs": probs, "complexity": cplx}
    if f"{p}.fused_experts.shared_expand.0.weight" in sd:            # DiversifiedExpertMoE (v0_14, :2499-2561)
        d = diversified_experts(sd, f"{p}.fused_experts", xd, w, idx, num_groups)
    elif f"{p}.fused_experts.shared_feature.0.weight" in sd:
        d = shared_inverted_experts(sd, f"{p}.fused_experts", xd, w, idx)
    else:
        d = plain_fused_experts(sd, f"{p}.fused_experts", xd, w, idx, num_experts, num_groups)
    cat = cross_path_gate(sd, f"{p}.cross_gate", s, d) if cross_gate else torch.cat([s, d], dim=1)
    oc = cat.shape[1]
    sg = shuffle_groups if oc % shuffle_groups == 0 else 1
    if sg > 1:
        cat = cat.view(B, sg, oc // sg, H, W).transpose(1, 2).reshape(B, oc, H, W)
    if f"{p}.refine_dw.0.weight" in sd:                                  # _apply_refine (gated.py:1947-1950)
        r = _gn(sd, f"{p}.refine_dw.1", _dw3(cat, sd[f"{p}.refine_dw.0.weight"]), num_groups)
        gg = F.silu(F.conv2d(F.adaptive_avg_pool2d(cat, 1), sd[f"{p}.refine_gate.1.weight"]))
        gg = torch.sigmoid(F.conv2d(gg, sd[f"{p}.refine_gate.3.weight"], sd[f"{p}.refine_gate.3.bias"]))
        cat = cat + torch.tanh(sd[f"{p}.refine_scale"]) * (r * gg)
    return _gn(sd, f"{p}.bn", F.conv2d(cat, sd[f"{p}.proj.weight"]), num_groups) + x
