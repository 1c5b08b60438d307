"""Config-5 mixture modules: `VisualEnhancedAdaptiveGateMoE`,
`C2fMoA` / `MoABlock`, `C2fMoT` / `MoTBlock`.

Drop-in boundary: the classes keep the reference's constructor signatures, parameter / buffer names, shapes and
registration order, so reference checkpoints of the v0_10 moa / mot YAMLs load unchanged and
`DetectionModel("yolo-master-moa-mot-n.yaml")` builds (tests/test_host_logic.py checks the key contract against keys
dumped from the real reference).

Host path: every module has its `_pack` (BN folds, grouped filters expanded to dense rows, channel padding to the
kernels' vector width, constant vectors for window-padding tokens, host scalars) and `_run` (NHWC dataflow over libymk entry
points, include/ymk_mixture.h).  There is no CPU or PyTorch fallback.  Validated on MI355X against the REAL reference's golden
vectors (tests/test_gpu_mixture.py); the dataflow alone is also checked on the CPU with the entry points emulated
(tests/emu_ops.py, test infrastructure) and with the kernel sources compiled for the host (tests/hostemu).

Reference: ultralytics/nn/modules/moe/gated.py:82-1764, moe/experts.py:183-269, moe/_gated_visual.py:32-75,
moa/{block,heads,router,wrappers}.py, mot/{block,experts,router,wrappers}.py.  Parameter containers are plain torch.nn
layers (memory only; never called).
"""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from .. import ops
from .modules import Conv, YmkModule, to_nhwc

def get_safe_groups(channels: int, desired_groups: int = 8) -> int:
    """Largest group count <= desired dividing channels (ultralytics/nn/modules/utils.py:108-115)."""
    if channels <= 0:
        return 1
    g = min(desired_groups, channels)
    while channels % g != 0:
        g -= 1
    return max(1, g)


def _gn(c, desired=8):
    return nn.GroupNorm(get_safe_groups(c, desired), c)


# ----------------------------------------------------------------------------------------- packing / run helpers
def _ceil(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _head_map(nh: int, hd: int, hdp: int, parts: int = 1):
    """Channel map for [part][head][hd] -> [part][head][hdp]: source index per destination channel, -1 = zero padding."""
    m = []
    for part in range(parts):
        for h in range(nh):
            m += [part * nh * hd + h * hd + d for d in range(hd)] + [-1] * (hdp - hd)
    return m


def _remap(w, cmap, dim: int):
    """Gather channels of `w` along `dim` by `cmap` (source index or -1 for a zero channel)."""
    if cmap is None:
        return w
    shape = list(w.shape)
    shape[dim] = len(cmap)
    out = w.new_zeros(shape)
    dst = torch.tensor([i for i, c in enumerate(cmap) if c >= 0], device=w.device)
    src = torch.tensor([c for c in cmap if c >= 0], device=w.device)
    out.index_copy_(dim, dst, w.index_select(dim, src))
    return out


def _pack_conv(conv, dtype, device, pad_cout_to=None, pad_cin_to=None, scale=None, rows=None, cols=None):
    """Bare nn.Conv2d (groups 1) or nn.Linear -> ([Cout][Kpad] in `dtype`, fp32 bias).  Zero padding of output / input
    channels keeps every operand inside the kernels' vector-width rules (`rows` / `cols`: channel maps with -1 for a zero
    channel, e.g. attention heads padded to a multiple of 8); `scale` [Cout] folds a per-channel factor (layer scale
    without a residual) into weights and bias."""
    w = conv.weight.detach().float().to(device)
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif conv.groups != 1:
        raise NotImplementedError("_pack_conv: dense convolutions only")
    b = conv.bias.detach().float().to(device) if conv.bias is not None else torch.zeros(w.shape[0], device=device)
    if scale is not None:
        sc = scale.detach().float().reshape(-1).to(device)
        w, b = w * sc.view(-1, 1, 1, 1), b * sc
    w, b = _remap(_remap(w, rows, 0), cols, 1), _remap(b, rows, 0)
    if pad_cout_to is not None and pad_cout_to > w.shape[0]:
        extra = pad_cout_to - w.shape[0]
        w = torch.cat([w, w.new_zeros((extra, *w.shape[1:]))], 0)
        b = torch.cat([b, b.new_zeros(extra)], 0)
    if pad_cin_to is not None and pad_cin_to > w.shape[1]:
        w = torch.cat([w, w.new_zeros((w.shape[0], pad_cin_to - w.shape[1], *w.shape[2:]))], 1)
    return ops.pack_conv_weight(w, dtype), b.contiguous()


def _pack_dw(conv, dtype, device, chans=None):
    if conv.groups != conv.in_channels or conv.in_channels != conv.out_channels or conv.bias is not None:
        raise NotImplementedError("_pack_dw: bias-free depthwise convolutions only")
    return ops.pack_dw_weight(_remap(conv.weight.detach().float().to(device), chans, 0), dtype)


def _pack_norm(norm, device):
    return (norm.weight.detach().float().to(device).contiguous(), norm.bias.detach().float().to(device).contiguous())


def _pack_bn_conv(m: Conv, dtype, device, scale=None):
    """modules.Conv (Conv2d + BN, 1x1 dense) with an optional per-channel factor folded in after the BN fold."""
    w, b = m._folded()
    w, b = w.to(device), b.to(device)
    if scale is not None:
        sc = scale.detach().float().reshape(-1).to(device)
        w, b = w * sc.view(-1, 1, 1, 1), b * sc
    return ops.pack_conv_weight(w, dtype), b.contiguous()


def _pack_conv_pair(a, b, dtype, device, pad_cout_to=None):
    """Two bare nn.Linear / 1x1 nn.Conv2d of the same input as ONE convolution, outputs side by side ([a | b], then zero padding)."""
    import types

    bias = None
    if a.bias is not None or b.bias is not None:
        za = a.bias if a.bias is not None else torch.zeros(a.weight.shape[0], device=a.weight.device)
        zb = b.bias if b.bias is not None else torch.zeros(b.weight.shape[0], device=b.weight.device)
        bias = torch.cat([za.detach().float(), zb.detach().float()], 0)
    both = types.SimpleNamespace(weight=torch.cat([a.weight.detach().float(), b.weight.detach().float()], 0), bias=bias, groups=1)
    return _pack_conv(both, dtype, device, pad_cout_to=pad_cout_to)


def _pack_bn_pair(a: Conv, b: Conv, dtype, device):
    """Two modules.Conv (1x1 + BN + SiLU) of the same input as ONE convolution with the outputs side by side."""
    (wa, ba), (wb, bb) = a._folded(), b._folded()
    return ops.pack_conv_weight(torch.cat([wa, wb], 0).to(device), dtype), torch.cat([ba, bb], 0).to(device).contiguous()


def _flat(p, device):
    return p.detach().float().reshape(-1).to(device).contiguous()


def _fold_ls(dtype) -> bool:
    """x + ls * conv(h): in the 16-bit modes the layer-scale factor is folded into the (activation-free) convolution's weights and the
    skip is that convolution's residual operand — one pass over the map less per residual (config 5: 76 launches, 1.7 ms per step).
    fp32 is the exact-parity configuration: it keeps the reference's order of operations, (conv(h)) * ls + x, so that NMS decisions
    on near-equal scores stay where the reference's are."""
    return dtype != torch.float32


def _ls_conv(h, pk, key, ls_key, res, out=None):
    """res + ls * conv(h) with pk[key] = (weights, bias): folded form when pk[ls_key] is None, scale-and-add pass otherwise."""
    if pk.get(ls_key) is None:
        return ops.conv2d(h, *pk[key], 1, 1, False, residual=res, out=out)
    return ops.scale_residual(ops.conv2d(h, *pk[key], 1, 1, False), pk[ls_key], res, out=out)


# ----------------------------------------------------------------------------------------- gated MoE
class DualStreamGateRouter(nn.Module):
    """moe/gated.py:82-122."""

    def __init__(self, in_channels, num_experts, top_k, temperature=1.0, local_reduction=16, pool_scale=4):
        super().__init__()
        self.num_experts, self.top_k = num_experts, top_k
        self.temperature = max(float(temperature), 1e-3)
        self.pool_scale = pool_scale
        self.global_fc = nn.Linear(2 * in_channels, num_experts, bias=False)
        reduced = max(in_channels // local_reduction, 4)
        self.local_conv = nn.Sequential(
            nn.Conv2d(in_channels, in_channels, 3, padding=1, groups=in_channels, bias=False), _gn(in_channels, 8), nn.SiLU(),
            nn.Conv2d(in_channels, reduced, 1, bias=False), _gn(reduced, 4), nn.SiLU(),
            nn.Conv2d(reduced, num_experts, 1, bias=True))
        self.alpha = nn.Parameter(torch.tensor(0.5))


class FusedExpertGroup(nn.Module):
    """moe/gated.py:1003-1036."""

    def __init__(self, in_channels, out_channels, num_experts, num_groups=8, top_k=2):
        super().__init__()
        fused = num_experts * out_channels
        g = min(get_safe_groups(in_channels, num_groups), fused)
        while g > 1 and (in_channels % g != 0 or fused % g != 0):
            g -= 1
        self.fused_conv = nn.Conv2d(in_channels, fused, 3, padding=1, groups=max(1, g), bias=False)
        self.expert_norm_weight = nn.Parameter(torch.ones(num_experts, out_channels))
        self.expert_norm_bias = nn.Parameter(torch.zeros(num_experts, out_channels))


class LowRankFusedExpertGroup(nn.Module):
    """moe/gated.py:1101-1142."""

    def __init__(self, in_channels, out_channels, num_experts, num_groups=8, top_k=2, bottleneck_ratio=0.5, min_channels=16):
        super().__init__()
        bc = min(in_channels, max(min_channels, int(round(in_channels * bottleneck_ratio))))
        self.bottleneck = nn.Sequential(nn.Conv2d(in_channels, bc, 1, bias=False), _gn(bc, num_groups), nn.SiLU())
        self.fused = FusedExpertGroup(bc, out_channels, num_experts, num_groups, top_k=top_k)


class SharedInvertedExpertGroup(nn.Module):
    """moe/experts.py:183-229."""

    def __init__(self, in_channels, out_channels, num_experts, expand_ratio=2.0, kernel_size=3, top_k=2, weight_threshold=0.0):
        super().__init__()
        hid = max(1, int(in_channels * expand_ratio))
        self.shared_feature = nn.Sequential(
            nn.Conv2d(in_channels, hid, 1, bias=False), _gn(hid), nn.SiLU(),
            nn.Conv2d(hid, hid, kernel_size, padding=kernel_size // 2, groups=hid, bias=False), _gn(hid), nn.SiLU())
        self.expert_projections = nn.ModuleList(
            nn.Sequential(nn.Conv2d(hid, out_channels, 1, bias=False), _gn(out_channels)) for _ in range(num_experts))


class VisualDetailGate(nn.Module):
    """moe/gated.py:1154-1169."""

    def __init__(self, channels, num_groups=8, reduction=8):
        super().__init__()
        hidden = max(channels // reduction, 8)
        self.detail_filter = nn.Sequential(
            nn.Conv2d(channels, channels, 3, padding=1, groups=channels, bias=False), _gn(channels, num_groups), nn.SiLU(),
            nn.Conv2d(channels, hidden, 1, bias=False), nn.SiLU(), nn.Conv2d(hidden, channels, 1, bias=True), nn.Sigmoid())
        self.detail_scale = nn.Parameter(torch.tensor(0.1))


class PyramidContextMixer(nn.Module):
    """moe/gated.py:1184-1207."""

    def __init__(self, channels, num_groups=8, pool_scales=(2, 4)):
        super().__init__()
        self.pool_scales = tuple(pool_scales)
        self.local_context = nn.Sequential(
            nn.Conv2d(channels, channels, 3, padding=1, groups=channels, bias=False), _gn(channels, num_groups), nn.SiLU())
        self.pool_projections = nn.ModuleList(
            nn.Sequential(nn.Conv2d(channels, channels, 1, bias=False), _gn(channels, num_groups), nn.SiLU())
            for _ in self.pool_scales)
        self.context_gate = nn.Sequential(nn.Conv2d(channels, channels, 1, bias=True), nn.Sigmoid())
        self.context_scale = nn.Parameter(torch.tensor(0.1))


class AdaptiveGateMoE(YmkModule):
    """v0_4 gated MoE and the base of the chain (moe/gated.py:268-640): SE-gated channel split, static DW+PW path, dual-stream
    router + batch-level complexity gate, routed experts, 1x1 projection + GroupNorm + x.  The later generations only swap the
    expert backend, add the channel shuffle and attach hooks (detail before routing; context / refine after the fusion), so they
    are this class with other settings:

        class (YAML generation)                         experts                          shuffle  hooks              forward
        AdaptiveGateMoE (v0_4)                          shared-inverted                  no       router_hooks=...   :530-580
        FusedAdaptiveGateMoE (v0_5)                     fused                            no       -                  (inherited)
        HybridAdaptiveGateMoE (v0_6)                    fused (E <= 8) / shared-inv.     yes      -                  :1340-1386
        HybridAdaptiveGateMoEv2 (v0_11)                 as v0_6, DualStreamGateRouterV2  yes      -                  (inherited)
        LowRankHybridAdaptiveGateMoE (v0_7)             low-rank fused / shared-inv.     yes      -                  (inherited)
        RefinedLowRankHybridAdaptiveGateMoE (v0_8)      "                                yes      refine             _gated_visual.py:32-75
        DetailAwareLowRankHybridAdaptiveGateMoE (v0_9)  "                                yes      detail             "
        ContextRefinedLowRankHybridAdaptiveGateMoE      "                                yes      context, refine    "
        VisualEnhancedAdaptiveGateMoE (v0_10)           "                                yes      detail, context, refine  "

    The complexity score reads the dynamic half before the hooks in AdaptiveGateMoE.forward (:548-552) and after the detail gate
    in `run_visual_hybrid_moe_forward` (_gated_visual.py:47-53): `_complexity_after_hooks`."""

    _backend_rule = "shared_inverted"     # "shared_inverted" | "fused" | "hybrid" | "low_rank_hybrid"
    _shuffle = False
    _router_v2 = False
    _hooks = ()
    _complexity_after_hooks = False
    _default_temperature = 1.0

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8,
                 initial_temperature=None, final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0,
                 entropy_loss_coeff=0.01, router_hooks=None, detail_reduction=8, refine_reduction=8, *, fused_expert_threshold=8,
                 shuffle_groups=2, bottleneck_ratio=0.5):
        super().__init__()
        if initial_temperature is None:
            initial_temperature = self._default_temperature
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_experts, self.top_k, self.num_groups = num_experts, top_k, num_groups
        self.initial_temperature, self.final_temperature = initial_temperature, final_temperature
        self.dynamic_channels = int(in_channels * split_ratio)
        self.static_channels = in_channels - self.dynamic_channels
        self.out_dynamic = int(out_channels * split_ratio)
        self.out_static = out_channels - self.out_dynamic
        self.shuffle_groups = (shuffle_groups if out_channels % shuffle_groups == 0 else 1) if self._shuffle else 1
        se_hidden = max(in_channels // 4, 4)
        self.se_gate = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(in_channels, se_hidden, bias=False),
                                     nn.SiLU(), nn.Linear(se_hidden, in_channels, bias=True), nn.Sigmoid())
        sc = self.static_channels
        self.static_net = nn.Sequential(
            nn.Conv2d(sc, sc, 3, padding=1, groups=sc, bias=False), nn.BatchNorm2d(sc), nn.SiLU(),
            nn.Conv2d(sc, self.out_static, 1, bias=False), nn.BatchNorm2d(self.out_static), nn.SiLU())
        router = DualStreamGateRouterV2 if self._router_v2 else DualStreamGateRouter
        self.routing = router(self.dynamic_channels, num_experts, top_k, temperature=initial_temperature)
        rule, few = self._backend_rule, num_experts <= fused_expert_threshold
        if rule == "fused" or (rule == "hybrid" and few):
            self.expert_backend = "fused"
            self.fused_experts = FusedExpertGroup(self.dynamic_channels, self.out_dynamic, num_experts, num_groups, top_k=top_k)
        elif rule == "low_rank_hybrid" and few:
            self.expert_backend = "low_rank_fused"
            self.fused_experts = LowRankFusedExpertGroup(self.dynamic_channels, self.out_dynamic, num_experts, num_groups,
                                                         top_k=top_k, bottleneck_ratio=bottleneck_ratio)
        else:
            self.expert_backend = "shared_inverted"
            self.fused_experts = SharedInvertedExpertGroup(self.dynamic_channels, self.out_dynamic, num_experts, top_k=top_k)
        self.complexity_estimator = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(self.dynamic_channels, 1, 1), nn.Sigmoid())
        self.proj = nn.Conv2d(out_channels, out_channels, 1, bias=False)
        self.bn = _gn(out_channels, num_groups)
        hooks = tuple(self._hooks)
        if router_hooks is not None:      # AdaptiveGateMoE(router_hooks=[...]) (gated.py:388-438): names in application order
            alias = {"feature_refinement": "refine"}
            names = [router_hooks] if isinstance(router_hooks, str) else list(router_hooks)
            hooks = tuple(alias.get(str(h).strip().lower(), str(h).strip().lower()) for h in names)
            if len(set(hooks)) != len(hooks) or any(h not in ("detail", "context", "refine") for h in hooks):
                raise KeyError(f"unknown or repeated router hook in {names}; available=['context', 'detail', 'feature_refinement', 'refine']")
        # learnable hook modules, in the reference's registration order of each class (state_dict key order)
        for h in self._hook_registration_order(hooks):
            if h == "refine":
                refine_hidden = max(out_channels // refine_reduction, 8)
                self.feature_refiner = nn.Sequential(
                    nn.Conv2d(out_channels, out_channels, 3, padding=1, groups=out_channels, bias=False), _gn(out_channels, num_groups),
                    nn.SiLU())
                self.feature_gate = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(out_channels, refine_hidden, 1, bias=False), nn.SiLU(),
                                                  nn.Conv2d(refine_hidden, out_channels, 1, bias=True), nn.Sigmoid())
                self.refine_scale = nn.Parameter(torch.tensor(0.1))
            elif h == "context":
                self.context_mixer = PyramidContextMixer(out_channels, num_groups)
            elif h == "detail":
                self.detail_gate = VisualDetailGate(self.dynamic_channels, num_groups, detail_reduction)
        self.router_hook_names = hooks

    def _hook_registration_order(self, hooks):
        if self._hooks:                      # subclasses: refine (v0_8 __init__) before context before detail (gated.py:1545-1560, 1693, 1751)
            return [h for h in ("refine", "context", "detail") if h in hooks]
        return [h for h in ("detail", "context", "refine") if h in hooks]    # configure_router_hooks (gated.py:415-435)

    # -- packing ---------------------------------------------------------------------------------------------------
    def _pack(self, dtype, device):
        import math

        f32 = torch.float32
        dyn = self.dynamic_channels
        se, rt = self.se_gate, self.routing
        if dyn % 8 or self.static_channels % 8 or self.out_dynamic % 8 or self.out_static % 8:
            raise NotImplementedError(f"ymk {type(self).__name__}: split {self.static_channels}+{dyn} breaks the 16-byte channel-vector rule")
        E4, red = _ceil(self.num_experts, 4), rt.local_conv[3].out_channels
        rp = _ceil(red, 4)
        sn = self.static_net
        dw_w, dw_b = ops.fold_bn(sn[0].weight.detach().float().to(device), sn[1].weight.float().to(device), sn[1].bias.float().to(device),
                                 sn[1].running_mean.float().to(device), sn[1].running_var.float().to(device), sn[1].eps)
        pw_w, pw_b = ops.fold_bn(sn[3].weight.detach().float().to(device), sn[4].weight.float().to(device), sn[4].bias.float().to(device),
                                 sn[4].running_mean.float().to(device), sn[4].running_var.float().to(device), sn[4].eps)
        gw, gb = _pack_conv(rt.global_fc, f32, device, pad_cout_to=E4)
        pk = {
            "se0": _pack_conv(se[2], f32, device), "se1": _pack_conv(se[4], f32, device),
            "st_dw": (ops.pack_dw_weight(dw_w, dtype), dw_b.contiguous()), "st_pw": (ops.pack_conv_weight(pw_w, dtype), pw_b.contiguous()),
            "cplx": _pack_conv(self.complexity_estimator[1], f32, device, pad_cout_to=4),
            "lc0": _pack_dw(rt.local_conv[0], f32, device), "lc1": _pack_norm(rt.local_conv[1], device),
            "lc3": _pack_conv(rt.local_conv[3], f32, device, pad_cout_to=rp), "lc4": _pack_norm(rt.local_conv[4], device),
            "lc_red": red, "lc_rp": rp,
            "lc6": _pack_conv(rt.local_conv[6], f32, device, pad_cout_to=E4, pad_cin_to=rp), "alpha": float(rt.alpha), "inv_temp": 1.0 / rt.temperature,
            "proj": _pack_conv(self.proj, dtype, device), "bn": _pack_norm(self.bn, device),
        }
        if self._router_v2:
            # expert_prior is added to the blended logits alpha * g + (1 - alpha) * l (gated.py:217-262): folded into the global
            # stream's bias as prior / alpha
            gb = gb.clone()
            gb[: self.num_experts] = rt.expert_prior.detach().float().to(device) / float(torch.sigmoid(rt.alpha.detach().float()))
            pk["sn"] = _pack_norm(rt.stat_norm, device)
        pk["gfc"] = (gw, gb.contiguous())
        if "detail" in self.router_hook_names:
            dg = self.detail_gate
            hp = torch.full((9, dyn), -1.0 / 9.0, device=device)      # x - avg_pool3x3(x) (zero padded, count_include_pad) as one stencil
            hp[4] += 1.0
            pk.update({"hp": hp.to(dtype).contiguous(),
                       "dg0": _pack_dw(dg.detail_filter[0], dtype, device), "dg1": _pack_norm(dg.detail_filter[1], device),
                       "dg3": _pack_conv(dg.detail_filter[3], dtype, device), "dg5": _pack_conv(dg.detail_filter[5], dtype, device),
                       "dg_s": math.tanh(float(dg.detail_scale))})
        if "context" in self.router_hook_names:
            cm = self.context_mixer
            pk.update({"cm0": _pack_dw(cm.local_context[0], dtype, device), "cm1": _pack_norm(cm.local_context[1], device),
                       "cmp": [(_pack_conv(q[0], dtype, device), _pack_norm(q[1], device)) for q in cm.pool_projections],
                       "cmg": _pack_conv(cm.context_gate[0], dtype, device), "cm_s": math.tanh(float(cm.context_scale))})
        if "refine" in self.router_hook_names:
            pk.update({"fr0": _pack_dw(self.feature_refiner[0], dtype, device), "fr1": _pack_norm(self.feature_refiner[1], device),
                       "fg1": _pack_conv(self.feature_gate[1], f32, device), "fg3": _pack_conv(self.feature_gate[3], f32, device),
                       "rf_s": math.tanh(float(self.refine_scale))})
        fe = self.fused_experts
        E, OC = self.num_experts, self.out_dynamic
        if self.expert_backend in ("low_rank_fused", "fused"):
            if self.expert_backend == "low_rank_fused":
                pk["bt0"] = _pack_conv(fe.bottleneck[0], dtype, device)
                pk["bt1"] = _pack_norm(fe.bottleneck[1], device)
                fe = fe.fused
            fc = fe.fused_conv
            w = fc.weight.detach().float().to(device)                       # [E*OC, cin/g, 3, 3], grouped
            cin, g = fc.in_channels, fc.groups
            cg, og = cin // g, (E * OC) // g
            dense = w.new_zeros((E * OC, cin, 3, 3))
            for grp in range(g):                                             # expand the grouped filter bank to dense rows:
                dense[grp * og:(grp + 1) * og, grp * cg:(grp + 1) * cg] = w[grp * og:(grp + 1) * og]   # only the routed experts' rows run
            pk["ew"] = ops.pack_conv_weight(dense, dtype).reshape(E, OC, -1).contiguous()
            pk["en"] = (fe.expert_norm_weight.detach().float().to(device).contiguous(),
                        fe.expert_norm_bias.detach().float().to(device).contiguous())
        else:
            sf = fe.shared_feature
            pk["sf0"], pk["sf1"] = _pack_conv(sf[0], dtype, device), _pack_norm(sf[1], device)
            pk["sf3"], pk["sf4"], pk["sf_k"] = _pack_dw(sf[3], dtype, device), _pack_norm(sf[4], device), sf[3].kernel_size[0]
            pk["ew"] = torch.stack([_pack_conv(q[0], dtype, device)[0] for q in fe.expert_projections]).contiguous()
            pk["en"] = (torch.stack([q[1].weight.detach().float() for q in fe.expert_projections]).to(device).contiguous(),
                        torch.stack([q[1].bias.detach().float() for q in fe.expert_projections]).to(device).contiguous())
        return pk

    # -- execution -------------------------------------------------------------------------------------------------
    def _run(self, x, out=None):
        """AdaptiveGateMoE.forward / HybridAdaptiveGateMoE.forward / run_visual_hybrid_moe_forward, eval (moe/gated.py:530-580,
        :1340-1386, moe/_gated_visual.py:32-75; pieces: gated.py:124-166 router, :217-262 router V2, :333-353 SE gate + static path,
        :455-492 complexity gate, :1058-1146 fused experts, :1171-1218 detail gate / context mixer, hooks.py:60-68 refinement;
        moe/experts.py:235-269 shared-inverted experts)."""
        B, H, W, C = x.shape
        pk = self._packed(x.device)
        st, dyn, ng, k = self.static_channels, self.dynamic_channels, self.num_groups, self.top_k
        gs = get_safe_groups
        hooks = self.router_hook_names
        # squeeze-excite gate over all channels, then the static / dynamic split
        gate = ops.conv2d_act(ops.conv2d(ops.channel_stats(x), *pk["se0"], 1, 1, True), *pk["se1"], 1, 1, "sigmoid")
        xg = ops.channel_gate(x, gate)
        xs, xd = xg[..., :st], xg[..., st:]
        cplx = None
        if not self._complexity_after_hooks:
            cplx = ops.conv2d(ops.channel_stats(xd), *pk["cplx"], 1, 1, False)[..., :1]
        if "detail" in hooks:   # detail gate on the dynamic half (pre-route hook)
            h = ops.dwconv2d(ops.dwconv2d(xd, pk["hp"], None, 3, False), pk["dg0"], None, 3, False)
            h = ops.conv2d(ops.group_norm(h, gs(dyn, ng), *pk["dg1"], 1e-5, act="silu"), *pk["dg3"], 1, 1, True)
            xd = ops.fma_gate(xd, xd, ops.conv2d_act(h, *pk["dg5"], 1, 1, "sigmoid"), pk["dg_s"])
        # static path: DW3x3+BN+SiLU -> 1x1+BN+SiLU (BN folded)
        s = ops.conv2d(ops.dwconv2d(xs, *pk["st_dw"], 3, True), *pk["st_pw"], 1, 1, True)
        # routing: global statistics stream + pooled local stream, decision tail with the batch-level complexity gate
        if cplx is None:
            cplx = ops.conv2d(ops.channel_stats(xd), *pk["cplx"], 1, 1, False)[..., :1]
        E = self.num_experts
        stats = ops.channel_stats(xd, want_std=True)
        if self._router_v2:
            stats = ops.layer_norm(stats, *pk["sn"], 1e-5)
        g_logits = ops.conv2d(stats, *pk["gfc"], 1, 1, False)[..., :E]
        ps = self.routing.pool_scale
        xl = ops.avg_pool(xd, ps if (H > ps and W > ps) else 1, out_dtype=torch.float32)
        h = ops.group_norm(ops.dwconv2d(xl, pk["lc0"], None, 3, False), gs(dyn, 8), *pk["lc1"], 1e-5, act="silu")
        h = ops.conv2d(h, *pk["lc3"], 1, 1, False)
        red, rp = pk["lc_red"], pk["lc_rp"]
        hn = h if red == rp else torch.zeros(h.shape, dtype=h.dtype, device=h.device)   # pad channels must stay zero
        ops.group_norm(h[..., :red], gs(red, 4), *pk["lc4"], 1e-5, act="silu", out=hn[..., :red])
        loc = ops.channel_stats(ops.conv2d(hn, *pk["lc6"], 1, 1, False))[..., :E]
        w, idx, probs, rows = ops.gated_route_decide(g_logits, loc, pk["alpha"], pk["inv_temp"], k, cplx)
        self.last_route = {"weights": w, "indices": idx, "probs": probs}   # rows: expert of image j*B + b in the slot-major expert batch
        # routed experts: only the selected experts' filter rows run
        OC = self.out_dynamic
        if self.expert_backend in ("low_rank_fused", "fused"):
            hb = xd
            if self.expert_backend == "low_rank_fused":
                hb = ops.group_norm(ops.conv2d(xd, *pk["bt0"], 1, 1, False), gs(pk["bt0"][0].shape[0], ng), *pk["bt1"], 1e-5, act="silu")
            f = ops.expert_conv(hb, pk["ew"], 3, idx)
            f = ops.group_norm(f, gs(OC, ng), *pk["en"], 1e-5, act="silu", affine_rows=rows)
        else:
            hs = ops.group_norm(ops.conv2d(xd, *pk["sf0"], 1, 1, False), gs(pk["sf0"][0].shape[0], 8), *pk["sf1"], 1e-5, act="silu")
            hs = ops.group_norm(ops.dwconv2d(hs, pk["sf3"], None, pk["sf_k"], False), gs(hs.shape[-1], 8), *pk["sf4"], 1e-5, act="silu")
            f = ops.group_norm(ops.expert_conv(hs, pk["ew"], 1, idx), gs(OC, 8), *pk["en"], 1e-5, affine_rows=rows)
        d = ops.weighted_sum(w, [f[j * B:(j + 1) * B] for j in range(k)])
        cat = ops.channel_shuffle_cat([s, d], self.shuffle_groups)
        oc = cat.shape[-1]
        for hook in hooks:      # post-fusion hooks in declaration order
            if hook == "context":   # pyramid context mixer
                ctx = [ops.group_norm(ops.dwconv2d(cat, pk["cm0"], None, 3, False), gs(oc, ng), *pk["cm1"], 1e-5, act="silu")]
                for (pw, pn), sc in zip(pk["cmp"], self.context_mixer.pool_scales):
                    hh, ww = max(1, H // sc), max(1, W // sc)
                    pooled = cat if (hh, ww) == (H, W) else ops.adaptive_avg_pool(cat, hh, ww)
                    ctx.append(ops.group_norm(ops.conv2d(pooled, *pw, 1, 1, False), gs(oc, ng), *pn, 1e-5, act="silu"))
                c = ops.mean_upsampled(ctx)
                cat = ops.fma_gate(cat, c, ops.conv2d_act(c, *pk["cmg"], 1, 1, "sigmoid"), pk["cm_s"])
            elif hook == "refine":  # feature refinement
                r = ops.group_norm(ops.dwconv2d(cat, pk["fr0"], None, 3, False), gs(oc, ng), *pk["fr1"], 1e-5, act="silu")
                g = ops.conv2d_act(ops.conv2d(ops.channel_stats(cat), *pk["fg1"], 1, 1, True), *pk["fg3"], 1, 1, "sigmoid")
                cat = ops.fma_gate(cat, r, g, pk["rf_s"])
        return ops.group_norm(ops.conv2d(cat, *pk["proj"], 1, 1, False), gs(oc, ng), *pk["bn"], 1e-5, residual=x, out=out)


class FusedAdaptiveGateMoE(AdaptiveGateMoE):
    """v0_5 (moe/gated.py:1232-1274): every expert in one grouped 3x3 filter bank (FusedExpertGroup), otherwise v0_4."""

    _backend_rule = "fused"

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.0,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff)


class HybridAdaptiveGateMoE(AdaptiveGateMoE):
    """v0_6 (moe/gated.py:1277-1386): fused experts up to `fused_expert_threshold`, shared-inverted above; channel shuffle of the
    concatenated paths before the projection."""

    _backend_rule = "hybrid"
    _shuffle = True
    _default_temperature = 1.2

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, **later):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold=fused_expert_threshold,
                         shuffle_groups=shuffle_groups, **later)
        self.fused_expert_threshold = fused_expert_threshold


class HybridAdaptiveGateMoEv2(HybridAdaptiveGateMoE):
    """v0_11 (moe/gated.py:1389-1452): v0_6 with DualStreamGateRouterV2 (LayerNorm on the statistics, learnable expert prior)."""

    _router_v2 = True


class LowRankHybridAdaptiveGateMoE(HybridAdaptiveGateMoE):
    """v0_7 (moe/gated.py:1455-1508): the fused backend behind a shared 1x1 bottleneck (LowRankFusedExpertGroup)."""

    _backend_rule = "low_rank_hybrid"

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, bottleneck_ratio=0.5, **later):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold, shuffle_groups,
                         bottleneck_ratio=bottleneck_ratio, **later)
        self.bottleneck_ratio = bottleneck_ratio


# Build-time expert-pool registry of SharedExpertMoE (moe/shared_expert_moe.py:27-29): pool_id -> {signature, "fused_experts": module}.
# parse_model clears it at model boundaries (nn/tasks.py:2037,2272); after construction the blocks keep the shared module as a child.
_SHARED_EXPERT_POOLS: dict = {}


class SharedExpertMoE(LowRankHybridAdaptiveGateMoE):
    """Cross-scale expert sharing (moe/shared_expert_moe.py:32-129; YAML cfg/models/master/v0_8/det/yolo-master-moe-mot-shared-n.yaml): a
    v0_7 block whose routed expert group is ONE module shared by every block built with the same `pool_id` — the first block of a pool
    owns it, later ones alias it (and must agree on dynamic channels, experts, top_k and bottleneck ratio: ValueError otherwise).  The
    `state_dict` lists the shared tensors under every member's prefix (torch lists shared modules per parent); `load_state_dict` writes
    them in module order, so the LAST member's entries are what all members compute with — exactly the reference's behaviour, obtained
    the same way (module aliasing).  Packing reads `self.fused_experts`, so every member packs the shared weights; nothing else differs
    from LowRankHybridAdaptiveGateMoE."""

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, bottleneck_ratio=0.5, pool_id="shared"):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold, shuffle_groups, bottleneck_ratio)
        self.pool_id = pool_id
        self._is_pool_owner = False
        self._setup_shared_pool()

    def _setup_shared_pool(self):
        """shared_expert_moe.py:85-115."""
        fe = self.fused_experts
        sig = {"in_channels": self.dynamic_channels, "out_channels": self.out_dynamic, "num_experts": getattr(fe, "num_experts", 0),
               "top_k": self.top_k, "bottleneck_ratio": self.bottleneck_ratio}
        pool = _SHARED_EXPERT_POOLS.get(self.pool_id)
        if pool is None:
            _SHARED_EXPERT_POOLS[self.pool_id] = {**sig, "fused_experts": fe}
            self._is_pool_owner = True
            return
        for k, v in sig.items():
            if k in pool and pool[k] != v:
                raise ValueError(f"SharedExpertMoE pool '{self.pool_id}' parameter mismatch: {k} expected {pool[k]}, got {v}. "
                                 "Blocks that share a pool must have the same channels/num_experts/top_k.")
        self.fused_experts = pool["fused_experts"]

    @classmethod
    def reset_shared_pools(cls):
        """Clear the build-time registry before / after constructing a model (shared_expert_moe.py:117-120)."""
        _SHARED_EXPERT_POOLS.clear()

    def get_pool_info(self):
        """shared_expert_moe.py:122-130."""
        return {"pool_id": self.pool_id, "is_owner": self._is_pool_owner, "num_experts": getattr(self.fused_experts, "num_experts", 0),
                "top_k": self.top_k, "dynamic_channels": self.dynamic_channels}


class RefinedLowRankHybridAdaptiveGateMoE(LowRankHybridAdaptiveGateMoE):
    """v0_8 (moe/gated.py:1511-1585): + gated residual depthwise refinement after the fusion."""

    _hooks = ("refine",)
    _complexity_after_hooks = True

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, bottleneck_ratio=0.5, refine_reduction=8, **later):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold, shuffle_groups,
                         bottleneck_ratio, refine_reduction=refine_reduction, **later)


class DetailAwareLowRankHybridAdaptiveGateMoE(LowRankHybridAdaptiveGateMoE):
    """v0_9 (moe/gated.py:1588-1642): + high-frequency detail gate on the dynamic half before routing."""

    _hooks = ("detail",)
    _complexity_after_hooks = True

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, bottleneck_ratio=0.5, detail_reduction=8):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold, shuffle_groups,
                         bottleneck_ratio, detail_reduction=detail_reduction)


class ContextRefinedLowRankHybridAdaptiveGateMoE(RefinedLowRankHybridAdaptiveGateMoE):
    """moe/gated.py:1645-1700: pyramid context mixer, then the v0_8 refinement."""

    _hooks = ("context", "refine")


class VisualEnhancedAdaptiveGateMoE(ContextRefinedLowRankHybridAdaptiveGateMoE):
    """v0_10, the end of the chain (moe/gated.py:1703-1764): detail gate before routing, context mixer and refinement after the
    fusion — the backbone block of the shipped MoA / MoT YAMLs (BASELINE config 5)."""

    _hooks = ("detail", "context", "refine")

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, bottleneck_ratio=0.5, refine_reduction=8, detail_reduction=8):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold, shuffle_groups,
                         bottleneck_ratio, refine_reduction, detail_reduction=detail_reduction)


class DualStreamGateRouterV2(DualStreamGateRouter):
    """moe/gated.py:181-215: LayerNorm over the [mean, std] statistics + a learnable per-expert prior on the logits."""

    def __init__(self, in_channels, num_experts, top_k, temperature=1.0, local_reduction=16, pool_scale=4, noise_std=0.1):
        super().__init__(in_channels, num_experts, top_k, temperature, local_reduction, pool_scale)
        self.stat_norm = nn.LayerNorm(2 * in_channels)
        self.expert_prior = nn.Parameter(torch.zeros(num_experts))
        self.register_buffer("_noise_progress", torch.tensor(0.0), persistent=False)


class CrossPathGate(nn.Module):
    """moe/gated.py:2347-2396 (parameters; the arithmetic is in GatedFusionMoE._run)."""

    def __init__(self, static_channels, dynamic_channels, out_channels, num_groups=8, drop_prob=0.1):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.static_channels, self.dynamic_channels, self.out_channels = static_channels, dynamic_channels, out_channels
        stat_dim = static_channels + dynamic_channels
        hidden = max(stat_dim // 4, 8)
        self.gate_net = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(stat_dim, hidden, bias=False), nn.SiLU(),
                                      nn.Linear(hidden, out_channels * 2, bias=True))
        self.gate_scale = nn.Parameter(torch.tensor(0.0))
        self.drop_scale = nn.Parameter(torch.tensor(1.0))


class OptimalHybridGateMoE(YmkModule):
    """v0_12 gated MoE (moe/gated.py:1846-2008; chain AdaptiveGateMoE :268 -> HybridAdaptiveGateMoE :1277 -> HybridAdaptiveGateMoEv2
    :1389): SE-gated channel split, static DW+PW path, DualStreamGateRouterV2 + batch-level complexity gate, fused (E <= 8) or
    shared-inverted experts, channel shuffle, residual DW refinement, 1x1 projection + GroupNorm + x.  Eval forward on libymk."""

    cross = False

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, refine=True, refine_reduction=8):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_experts, self.top_k, self.num_groups = num_experts, top_k, num_groups
        self.initial_temperature, self.final_temperature = initial_temperature, final_temperature
        self.dynamic_channels = int(in_channels * split_ratio)
        self.static_channels = in_channels - self.dynamic_channels
        self.out_dynamic = int(out_channels * split_ratio)
        self.out_static = out_channels - self.out_dynamic
        self.shuffle_groups = shuffle_groups if out_channels % shuffle_groups == 0 else 1
        se_hidden = max(in_channels // 4, 4)
        self.se_gate = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(in_channels, se_hidden, bias=False),
                                     nn.SiLU(), nn.Linear(se_hidden, in_channels, bias=True), nn.Sigmoid())
        sc = self.static_channels
        self.static_net = nn.Sequential(
            nn.Conv2d(sc, sc, 3, padding=1, groups=sc, bias=False), nn.BatchNorm2d(sc), nn.SiLU(),
            nn.Conv2d(sc, self.out_static, 1, bias=False), nn.BatchNorm2d(self.out_static), nn.SiLU())
        self.routing = DualStreamGateRouterV2(self.dynamic_channels, num_experts, top_k, temperature=initial_temperature)
        if num_experts <= fused_expert_threshold:
            self.expert_backend = "fused"
            self.fused_experts = FusedExpertGroup(self.dynamic_channels, self.out_dynamic, num_experts, num_groups, top_k=top_k)
        else:
            self.expert_backend = "shared_inverted"
            self.fused_experts = SharedInvertedExpertGroup(self.dynamic_channels, self.out_dynamic, num_experts, top_k=top_k)
        self.complexity_estimator = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(self.dynamic_channels, 1, 1), nn.Sigmoid())
        self.proj = nn.Conv2d(out_channels, out_channels, 1, bias=False)
        self.bn = _gn(out_channels, num_groups)
        self.refine = refine
        if refine:
            refine_hidden = max(out_channels // refine_reduction, 8)
            self.refine_dw = nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, padding=1, groups=out_channels, bias=False),
                                           _gn(out_channels, num_groups))
            self.refine_gate = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(out_channels, refine_hidden, 1, bias=False), nn.SiLU(),
                                             nn.Conv2d(refine_hidden, out_channels, 1, bias=True), nn.Sigmoid())
            self.refine_scale = nn.Parameter(torch.tensor(0.1))

    def _pack(self, dtype, device):
        import math

        f32 = torch.float32
        dyn = self.dynamic_channels
        se, rt = self.se_gate, self.routing
        if dyn % 8 or self.static_channels % 8 or self.out_dynamic % 8 or self.out_static % 8:
            raise NotImplementedError(f"ymk {type(self).__name__}: split {self.static_channels}+{dyn} breaks the 16-byte channel-vector rule")
        E4, red = _ceil(self.num_experts, 4), rt.local_conv[3].out_channels
        rp = _ceil(red, 4)
        sn = self.static_net
        dw_w, dw_b = ops.fold_bn(sn[0].weight.detach().float().to(device), sn[1].weight.float().to(device), sn[1].bias.float().to(device),
                                 sn[1].running_mean.float().to(device), sn[1].running_var.float().to(device), sn[1].eps)
        pw_w, pw_b = ops.fold_bn(sn[3].weight.detach().float().to(device), sn[4].weight.float().to(device), sn[4].bias.float().to(device),
                                 sn[4].running_mean.float().to(device), sn[4].running_var.float().to(device), sn[4].eps)
        alpha = float(torch.sigmoid(rt.alpha.detach().float()))
        lin = nn.Linear(2 * dyn, self.num_experts, bias=False)
        lin.weight = nn.Parameter(self._global_weight(rt).detach().float(), requires_grad=False)
        gw, gb = _pack_conv(lin, f32, device, pad_cout_to=E4)
        # expert_prior is added to the blended logits alpha * g + (1 - alpha) * l: folded into the global stream as prior / alpha
        gb = gb.clone()
        gb[: self.num_experts] = rt.expert_prior.detach().float().to(device) / alpha
        pk = {
            "se0": _pack_conv(se[2], f32, device), "se1": _pack_conv(se[4], f32, device),
            "st_dw": (ops.pack_dw_weight(dw_w, dtype), dw_b.contiguous()), "st_pw": (ops.pack_conv_weight(pw_w, dtype), pw_b.contiguous()),
            "cplx": _pack_conv(self.complexity_estimator[1], f32, device, pad_cout_to=4),
            "sn": _pack_norm(rt.stat_norm, device), "gfc": (gw, gb.contiguous()),
            "lc0": _pack_dw(rt.local_conv[0], f32, device), "lc1": _pack_norm(rt.local_conv[1], device),
            "lc3": _pack_conv(rt.local_conv[3], f32, device, pad_cout_to=rp), "lc4": _pack_norm(rt.local_conv[4], device),
            "lc_red": red, "lc_rp": rp,
            "lc6": _pack_conv(rt.local_conv[6], f32, device, pad_cout_to=E4, pad_cin_to=rp), "alpha": float(rt.alpha), "inv_temp": 1.0 / rt.temperature,
            "proj": _pack_conv(self.proj, dtype, device), "bn": _pack_norm(self.bn, device),
        }
        if self.refine:
            pk.update({"rd0": _pack_dw(self.refine_dw[0], dtype, device), "rd1": _pack_norm(self.refine_dw[1], device),
                       "rg1": _pack_conv(self.refine_gate[1], f32, device), "rg3": _pack_conv(self.refine_gate[3], f32, device),
                       "rf_s": math.tanh(float(self.refine_scale))})
        fe = self.fused_experts
        E, OC = self.num_experts, self.out_dynamic
        if self.expert_backend == "fused":
            fc = fe.fused_conv
            w = fc.weight.detach().float().to(device)                       # [E*OC, dyn/g, 3, 3], grouped
            cin, g = fc.in_channels, fc.groups
            cg, og = cin // g, (E * OC) // g
            dense = w.new_zeros((E * OC, cin, 3, 3))
            for grp in range(g):                                             # grouped filter bank -> dense rows (only routed rows run)
                dense[grp * og:(grp + 1) * og, grp * cg:(grp + 1) * cg] = w[grp * og:(grp + 1) * og]
            pk["ew"] = ops.pack_conv_weight(dense, dtype).reshape(E, OC, -1).contiguous()
            pk["en"] = (fe.expert_norm_weight.detach().float().to(device).contiguous(), fe.expert_norm_bias.detach().float().to(device).contiguous())
        elif self.expert_backend == "diversified":
            pk["sx0"], pk["sx1"] = _pack_conv(fe.shared_expand[0], dtype, device), _pack_norm(fe.shared_expand[1], device)
            pk["dww"] = torch.stack([_pack_dw(q[0], dtype, device) for q in fe.dw_layers]).contiguous()           # [E][9][hidden]
            pk["dwd"] = torch.tensor([q[0].dilation[0] for q in fe.dw_layers], dtype=torch.int32, device=device)
            pk["dwn"] = (torch.stack([q[1].weight.detach().float() for q in fe.dw_layers]).to(device).contiguous(),
                         torch.stack([q[1].bias.detach().float() for q in fe.dw_layers]).to(device).contiguous())
            pk["ew"] = torch.stack([_pack_conv(q[0], dtype, device)[0] for q in fe.expert_projections]).contiguous()
            pk["en"] = (torch.stack([q[1].weight.detach().float() for q in fe.expert_projections]).to(device).contiguous(),
                        torch.stack([q[1].bias.detach().float() for q in fe.expert_projections]).to(device).contiguous())
        else:
            sf = fe.shared_feature
            pk["sf0"], pk["sf1"] = _pack_conv(sf[0], dtype, device), _pack_norm(sf[1], device)
            pk["sf3"], pk["sf4"], pk["sf_k"] = _pack_dw(sf[3], dtype, device), _pack_norm(sf[4], device), sf[3].kernel_size[0]
            pk["ew"] = torch.stack([_pack_conv(p[0], dtype, device)[0] for p in fe.expert_projections]).contiguous()
            pk["en"] = (torch.stack([p[1].weight.detach().float() for p in fe.expert_projections]).to(device).contiguous(),
                        torch.stack([p[1].bias.detach().float() for p in fe.expert_projections]).to(device).contiguous())
        if self.cross:
            cg_ = self.cross_gate
            oc = self.out_static + self.out_dynamic
            c = 0.5 * math.tanh(float(cg_.gate_scale))
            # gate = 0.5 + c * sigmoid(raw): an affine map of the sigmoid, applied as a diagonal 1x1 convolution over its first oc entries
            diag = torch.zeros((oc, oc, 1, 1), device=device)
            diag[torch.arange(oc), torch.arange(oc), 0, 0] = c
            pk.update({"cg0": _pack_conv(cg_.gate_net[2], f32, device), "cg1": _pack_conv(cg_.gate_net[4], f32, device),
                       "cg_aff": (ops.pack_conv_weight(diag, f32), torch.full((oc,), 0.5, device=device))})
        return pk

    def _global_weight(self, rt):
        """[E][2 * dyn] matrix of the router's global stream over the normalised statistics."""
        return rt.global_fc.weight

    def _fuse_paths(self, s, d, pk):
        """[static | dynamic] -> channel-shuffled concatenation (gated.py:1333-1338)."""
        return ops.channel_shuffle_cat([s, d], self.shuffle_groups)

    def _run(self, x, out=None):
        B, H, W, C = x.shape
        pk = self._packed(x.device)
        st, dyn, ng, k = self.static_channels, self.dynamic_channels, self.num_groups, self.top_k
        gs = get_safe_groups
        gate = ops.conv2d_act(ops.conv2d(ops.channel_stats(x), *pk["se0"], 1, 1, True), *pk["se1"], 1, 1, "sigmoid")
        xg = ops.channel_gate(x, gate)
        xs, xd = xg[..., :st], xg[..., st:]
        s = ops.conv2d(ops.dwconv2d(xs, *pk["st_dw"], 3, True), *pk["st_pw"], 1, 1, True)
        cplx = ops.conv2d(ops.channel_stats(xd), *pk["cplx"], 1, 1, False)[..., :1]
        E = self.num_experts
        stats = ops.layer_norm(ops.channel_stats(xd, want_std=True), *pk["sn"], 1e-5)
        g_logits = ops.conv2d(stats, *pk["gfc"], 1, 1, False)[..., :E]
        ps = self.routing.pool_scale
        xl = ops.avg_pool(xd, ps if (H > ps and W > ps) else 1, out_dtype=torch.float32)
        h = ops.group_norm(ops.dwconv2d(xl, pk["lc0"], None, 3, False), gs(dyn, 8), *pk["lc1"], 1e-5, act="silu")
        h = ops.conv2d(h, *pk["lc3"], 1, 1, False)
        red, rp = pk["lc_red"], pk["lc_rp"]
        hn = h if red == rp else torch.zeros(h.shape, dtype=h.dtype, device=h.device)
        ops.group_norm(h[..., :red], gs(red, 4), *pk["lc4"], 1e-5, act="silu", out=hn[..., :red])
        loc = ops.channel_stats(ops.conv2d(hn, *pk["lc6"], 1, 1, False))[..., :E]
        w, idx, probs, rows = ops.gated_route_decide(g_logits, loc, pk["alpha"], pk["inv_temp"], k, cplx)
        self.last_route = {"weights": w, "indices": idx, "probs": probs}
        OC = self.out_dynamic
        if self.expert_backend == "fused":
            f = ops.expert_conv(xd, pk["ew"], 3, idx)
            f = ops.group_norm(f, gs(OC, ng), *pk["en"], 1e-5, act="silu", affine_rows=rows)
        elif self.expert_backend == "diversified":   # DiversifiedExpertGroup (gated.py:2296-2330): per-expert dilated DW3x3 between expand and projection
            hs = ops.group_norm(ops.conv2d(xd, *pk["sx0"], 1, 1, False), gs(pk["sx0"][0].shape[0], ng), *pk["sx1"], 1e-5, act="silu")
            hd = ops.expert_dw3(hs, pk["dww"], pk["dwd"], idx)                                                   # [k * B, H, W, hidden], slot-major
            hd = ops.group_norm(hd, gs(hd.shape[-1], ng), *pk["dwn"], 1e-5, act="silu", affine_rows=rows)
            f = ops.expert_conv(hd, pk["ew"], 1, rows.view(-1, 1).contiguous())                                 # image n of the slot-major batch -> its own expert
            f = ops.group_norm(f, gs(OC, ng), *pk["en"], 1e-5, affine_rows=rows)
        else:
            hs = ops.group_norm(ops.conv2d(xd, *pk["sf0"], 1, 1, False), gs(pk["sf0"][0].shape[0], 8), *pk["sf1"], 1e-5, act="silu")
            hs = ops.group_norm(ops.dwconv2d(hs, pk["sf3"], None, pk["sf_k"], False), gs(hs.shape[-1], 8), *pk["sf4"], 1e-5, act="silu")
            f = ops.group_norm(ops.expert_conv(hs, pk["ew"], 1, idx), gs(OC, 8), *pk["en"], 1e-5, affine_rows=rows)
        d = ops.weighted_sum(w, [f[j * B:(j + 1) * B] for j in range(k)])
        cat = self._fuse_paths(s, d, pk)
        oc = cat.shape[-1]
        if self.refine:   # x + tanh(scale) * GN(DW3x3(x)) * SE(x)   (gated.py:1947-1950)
            r = ops.group_norm(ops.dwconv2d(cat, pk["rd0"], None, 3, False), gs(oc, ng), *pk["rd1"], 1e-5)
            g = ops.conv2d_act(ops.conv2d(ops.channel_stats(cat), *pk["rg1"], 1, 1, True), *pk["rg3"], 1, 1, "sigmoid")
            cat = ops.fma_gate(cat, r, g, pk["rf_s"])
        return ops.group_norm(ops.conv2d(cat, *pk["proj"], 1, 1, False), gs(oc, ng), *pk["bn"], 1e-5, residual=x, out=out)


class GatedFusionMoE(OptimalHybridGateMoE):
    """v0_15 gated MoE (moe/gated.py:2564-2693): OptimalHybridGateMoE with a content-aware CrossPathGate between the two paths and
    the channel shuffle (stochastic depth is training-only)."""

    cross = True

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, refine=True, refine_reduction=8, drop_prob=0.05):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold, shuffle_groups, refine,
                         refine_reduction)
        self.cross_gate = CrossPathGate(self.out_static, self.out_dynamic, out_channels, num_groups=num_groups, drop_prob=drop_prob)

    def _fuse_paths(self, s, d, pk):
        """CrossPathGate.forward (gated.py:2398-2428): gate = 0.5 + tanh(scale) / 2 * sigmoid(MLP(GAP [s | d])) per image and channel,
        applied to both paths before the shuffle."""
        cs, cd = self.out_static, self.out_dynamic
        B = s.shape[0]
        stats = torch.empty((B, 1, 1, cs + cd), dtype=torch.float32, device=s.device)
        ops.copy_channels(ops.channel_stats(s), stats[..., :cs])
        ops.copy_channels(ops.channel_stats(d), stats[..., cs:])
        sig = ops.conv2d_act(ops.conv2d(stats, *pk["cg0"], 1, 1, True), *pk["cg1"], 1, 1, "sigmoid")     # [B,1,1,2*out_channels]
        gate = ops.conv2d(sig[..., : cs + cd], *pk["cg_aff"], 1, 1, False)
        gs_, gd_ = gate[..., :cs].contiguous(), gate[..., cs:].contiguous()     # channel_gate takes dense [B,1,1,C] gates (two tiny copies)
        return ops.channel_shuffle_cat([ops.channel_gate(s, gs_), ops.channel_gate(d, gd_)], self.shuffle_groups)


class MultiHeadRouterV3(nn.Module):
    """moe/gated.py:2026-2106 (parameters; the arithmetic is folded at pack time, see MultiHeadRouterMoE)."""

    def __init__(self, in_channels, num_experts, top_k, temperature=1.0, num_heads=4, local_reduction=16, pool_scale=4, noise_std=0.1,
                 expert_dropout=0.1):
        super().__init__()
        self.num_experts, self.top_k = num_experts, top_k
        self.temperature = max(float(temperature), 1e-3)
        self.pool_scale = pool_scale
        self.num_heads = max(1, min(num_heads, num_experts))
        stat_dim = 2 * in_channels
        self.stat_norm = nn.LayerNorm(stat_dim)
        self._head_dim = max(stat_dim // self.num_heads, 4)
        self.heads = nn.ModuleList([nn.Linear(self._head_dim, num_experts, bias=False) for _ in range(self.num_heads)])
        self.global_proj = nn.Linear(stat_dim, num_experts, bias=False)
        self.head_alpha = nn.Parameter(torch.ones(self.num_heads) / self.num_heads)
        self.global_weight = nn.Parameter(torch.tensor(0.1))
        self.expert_prior = nn.Parameter(torch.zeros(num_experts))
        reduced = max(in_channels // local_reduction, 4)
        self.local_conv = nn.Sequential(
            nn.Conv2d(in_channels, in_channels, 3, padding=1, groups=in_channels, bias=False), _gn(in_channels, 8), nn.SiLU(),
            nn.Conv2d(in_channels, reduced, 1, bias=False), _gn(reduced, 4), nn.SiLU(), nn.Conv2d(reduced, num_experts, 1, bias=True))
        self.alpha = nn.Parameter(torch.tensor(0.5))
        self.register_buffer("_noise_progress", torch.tensor(0.0), persistent=False)


class MultiHeadRouterMoE(OptimalHybridGateMoE):
    """v0_13 gated MoE (moe/gated.py:2430-2496): OptimalHybridGateMoE routed by MultiHeadRouterV3 (:2108-2190).  Every head and the
    full-width projection are linear in the same normalised statistics vector, so the router's global stream is ONE [E][2 * dyn]
    matrix, built at pack time: sigmoid(global_weight) * W_global + (1 - sigmoid(global_weight)) * sum_i hw_i * W_i placed on head
    i's slice (hw = sigmoid(head_alpha) / (sum + 1e-6); slices beyond the statistics are padding, statistics beyond the slices are
    not read by any head).  Everything after it is the v0_12 path."""

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, refine=True, refine_reduction=8, num_heads=4, expert_dropout=0.05):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold, shuffle_groups, refine,
                         refine_reduction)
        self.routing = MultiHeadRouterV3(self.dynamic_channels, num_experts, top_k, temperature=initial_temperature, num_heads=num_heads,
                                         expert_dropout=expert_dropout)

    def _global_weight(self, rt):
        gw = torch.sigmoid(rt.global_weight.detach().float())
        hw = torch.sigmoid(rt.head_alpha.detach().float())
        hw = hw / (hw.sum() + 1e-6)
        w = gw * rt.global_proj.weight.detach().float()
        sd_, hd = w.shape[1], rt._head_dim
        for i, h in enumerate(rt.heads):
            lo, hi = i * hd, min((i + 1) * hd, sd_)
            if lo < hi:
                w[:, lo:hi] += (1 - gw) * hw[i] * h.weight.detach().float()[:, : hi - lo]
        return w


class DiversifiedExpertGroup(nn.Module):
    """moe/gated.py:2214-2294 (parameters; the arithmetic is in OptimalHybridGateMoE._run, backend "diversified")."""

    def __init__(self, in_channels, out_channels, num_experts, expand_ratio=2.0, top_k=2, weight_threshold=0.0, num_groups=8):
        super().__init__()
        self.in_channels, self.out_channels, self.num_experts, self.top_k = in_channels, out_channels, num_experts, top_k
        hidden = max(1, int(in_channels * expand_ratio))
        self.shared_expand = nn.Sequential(nn.Conv2d(in_channels, hidden, 1, bias=False), _gn(hidden, num_groups), nn.SiLU())
        self.dw_layers = nn.ModuleList()
        self.dw_dilations = nn.ParameterList()
        for i in range(num_experts):
            d = 1 + (i // 2)
            self.dw_layers.append(nn.Sequential(nn.Conv2d(hidden, hidden, 3, padding=d, dilation=d, groups=hidden, bias=False),
                                                _gn(hidden, num_groups), nn.SiLU()))
            self.dw_dilations.append(nn.Parameter(torch.tensor(float(d))))
        self.expert_projections = nn.ModuleList(nn.Sequential(nn.Conv2d(hidden, out_channels, 1, bias=False), _gn(out_channels, num_groups))
                                                for _ in range(num_experts))


class DiversifiedExpertMoE(OptimalHybridGateMoE):
    """v0_14 gated MoE (moe/gated.py:2499-2561): OptimalHybridGateMoE whose experts are a shared 1x1 expansion, a per-expert DILATED
    depthwise 3x3 (dilation 1 + i // 2) and a per-expert 1x1 projection, for every expert count.  The depthwise stage runs per routed
    (image, slot) pair with the expert's own filter and dilation (ymk_expert_dw3); the projection is the sparse expert convolution over
    the slot-major batch."""

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, initial_temperature=1.2,
                 final_temperature=0.5, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, entropy_loss_coeff=0.01,
                 fused_expert_threshold=8, shuffle_groups=2, refine=True, refine_reduction=8):
        super().__init__(in_channels, out_channels, num_experts, top_k, split_ratio, num_groups, initial_temperature, final_temperature,
                         balance_loss_coeff, router_z_loss_coeff, entropy_loss_coeff, fused_expert_threshold, shuffle_groups, refine,
                         refine_reduction)
        self.expert_backend = "diversified"
        self.fused_experts = DiversifiedExpertGroup(self.dynamic_channels, self.out_dynamic, num_experts, expand_ratio=2.0, top_k=top_k,
                                                    weight_threshold=0.0, num_groups=num_groups)


# ----------------------------------------------------------------------------------------- MoA
class _MoARouter(nn.Module):
    """moa/router.py:29-48."""

    def __init__(self, dim, num_groups, reduction=8, temperature=1.0):
        super().__init__()
        self.temperature = max(temperature, 0.1)
        hidden = max(dim // reduction, num_groups * 2)
        self.router = nn.Sequential(nn.Conv2d(dim, hidden, 1, bias=False), _gn(hidden, 4), nn.SiLU(),
                                    nn.Conv2d(hidden, num_groups, 1, bias=True))


class _LocalAttnHead(nn.Module):
    """moa/heads.py:120-141."""

    def __init__(self, dim, num_heads, head_dim=None, window_size=7):
        super().__init__()
        self.num_heads, self.head_dim, self.window_size = num_heads, head_dim or max(dim // num_heads, 16), max(1, int(window_size))
        inner = self.head_dim * num_heads
        self.qkv_dw = nn.Conv2d(dim, dim, 3, padding=1, groups=dim, bias=False)
        self.qkv_pw = nn.Conv2d(dim, inner * 3, 1, bias=False)
        self.proj = nn.Conv2d(inner, dim, 1, bias=False)
        self.pe = nn.Conv2d(inner, inner, 7, padding=3, groups=inner, bias=False)
        self.norm = _gn(dim)


class _RegionalAttnHead(nn.Module):
    """moa/heads.py:166-206."""

    def __init__(self, dim, num_heads, head_dim=None, pool_stride=2, max_kv_tokens=4096):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, head_dim or max(dim // num_heads, 16)
        self.pool_stride, self.max_kv_tokens = pool_stride, max_kv_tokens
        inner = self.head_dim * num_heads
        self.q_proj = nn.Conv2d(dim, inner, 1, bias=False)
        self.kv_proj = nn.Conv2d(dim, inner * 2, 1, bias=False)
        self.proj = nn.Conv2d(inner, dim, 1, bias=False)
        self.norm = _gn(dim)


class _GlobalAttnHead(nn.Module):
    """moa/heads.py:256-312: the orthogonal random-feature basis is a persistent buffer seeded per block."""

    def __init__(self, dim, num_heads, head_dim=None, nb_features=64, rf_seed=0x5F3759DF):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, head_dim or max(dim // num_heads, 16)
        inner = self.head_dim * num_heads
        self.qkv = nn.Conv2d(dim, inner * 3, 1, bias=False)
        self.proj = nn.Conv2d(inner, dim, 1, bias=False)
        self.norm = _gn(dim)
        eff = min(nb_features, self.head_dim)
        with torch.no_grad():
            gen = torch.Generator().manual_seed(rf_seed)
            rf, _ = torch.linalg.qr(torch.randn(self.head_dim, self.head_dim, generator=gen, dtype=torch.float32))
        self.register_buffer("_rf_matrix", rf[:eff].contiguous(), persistent=True)


class MoABlock(YmkModule):
    """moa/block.py:21-278 (eval, dense soft routing).  Host orchestration over libymk entry points (include/ymk_mixture.h);
    checked on MI355X (tests/test_gpu_mixture.py) and, dataflow only, on the CPU (tests/test_host_mixture.py) against the real
    reference's golden vectors."""

    NUM_GROUPS = 3

    def __init__(self, dim, num_heads=8, mlp_ratio=2.0, temperature=1.0, attn_drop=0.0, shortcut=True, aux_loss_coeff=0.01,
                 block_index=0, local_window_size=7, sequential_heads=True, regional_max_kv_tokens=4096,
                 sparse_inference=False, sparse_inference_threshold=0.02, inference_sparse_threshold=None):
        super().__init__()
        if num_heads <= 0 or num_heads % self.NUM_GROUPS != 0:
            raise ValueError(f"num_heads ({num_heads}) must be positive and divisible by NUM_GROUPS ({self.NUM_GROUPS})")
        if inference_sparse_threshold is not None:    # the older spelling of the same option switches it on (moa/block.py:67-76)
            if sparse_inference_threshold != 0.02 and sparse_inference_threshold != inference_sparse_threshold:
                raise ValueError("Specify only one sparse inference threshold: sparse_inference_threshold or inference_sparse_threshold.")
            sparse_inference_threshold, sparse_inference = inference_sparse_threshold, True
        self.sparse_inference = bool(sparse_inference)
        self.sparse_inference_threshold = float(sparse_inference_threshold)
        if not 0.0 <= self.sparse_inference_threshold < 1.0:
            raise ValueError("sparse_inference_threshold must be in [0, 1)")
        self.dim, self.shortcut = dim, shortcut
        head_dim = max(dim // num_heads, 16)
        hpg = num_heads // self.NUM_GROUPS
        ls = torch.ones(dim, 1, 1) * (0.1 if shortcut else 1.0)
        self.ls_attn = nn.Parameter(ls.clone())
        self.ls_ffn = nn.Parameter(ls.clone())
        self.local_head = _LocalAttnHead(dim, hpg, head_dim, window_size=local_window_size)
        self.region_head = _RegionalAttnHead(dim, hpg, head_dim, max_kv_tokens=regional_max_kv_tokens)
        self.global_head = _GlobalAttnHead(dim, hpg, head_dim, rf_seed=block_index * 7919 + 2 * 65537)
        self.router = _MoARouter(dim, self.NUM_GROUPS, temperature=temperature)
        self.fusion = Conv(dim, dim, 1, act=False)
        hidden = int(dim * mlp_ratio)
        self.ffn = nn.Sequential(Conv(dim, hidden, 1), Conv(hidden, dim, 1, act=False))

    # linear-attention switch-over of the global head (moa/_constants.py)
    LINEAR_ATTN_THRESHOLD = 512
    LINEAR_ATTN_BLEND_WINDOW = 64

    def _pack(self, dtype, device):
        f32 = torch.float32
        # head_dim = max(dim // num_heads, 16) need not be a multiple of the 16-byte channel vector (21 at the L scale of
        # BASELINE config 5): every head is zero-padded to hdp channels in the packed q / k / v / pe / proj weights and in
        # the random-feature basis, which changes no dot product and leaves the padded output channels at zero
        nh, hd = self.local_head.num_heads, self.local_head.head_dim
        hdp = _ceil(hd, 8)
        m1, m2, m3 = (None, None, None) if hdp == hd else (_head_map(nh, hd, hdp, 1), _head_map(nh, hd, hdp, 2), _head_map(nh, hd, hdp, 3))
        r = self.router.router
        hid = r[0].out_channels
        hp = _ceil(hid, 4)
        lh, rh, gh = self.local_head, self.region_head, self.global_head
        pk = {
            # router: first 1x1 reads the block input (compute dtype) and writes fp32; the rest stays fp32 (router.py:55-61)
            "r0": _pack_conv(r[0], dtype, device, pad_cout_to=hp), "r1": _pack_norm(r[1], device), "r_hid": hid, "r_hp": hp,
            "r3": _pack_conv(r[3], f32, device, pad_cout_to=4, pad_cin_to=hp),
            "hdp": hdp,
            "l_dw": _pack_dw(lh.qkv_dw, dtype, device), "l_qkv": _pack_conv(lh.qkv_pw, dtype, device, rows=m3),
            "l_pe": _pack_dw(lh.pe, dtype, device, chans=m1), "l_proj": _pack_conv(lh.proj, dtype, device, cols=m1),
            "l_norm": _pack_norm(lh.norm, device),
            "g_q": _pack_conv(rh.q_proj, dtype, device, rows=m1), "g_kv": _pack_conv(rh.kv_proj, dtype, device, rows=m2),
            "g_proj": _pack_conv(rh.proj, dtype, device, cols=m1), "g_norm": _pack_norm(rh.norm, device),
            "a_qkv": _pack_conv(gh.qkv, dtype, device, rows=m3), "a_proj": _pack_conv(gh.proj, dtype, device, cols=m1),
            "a_norm": _pack_norm(gh.norm, device),
            "rf": _remap(gh._rf_matrix.detach().float().to(device), None if hdp == hd else list(range(hd)) + [-1] * (hdp - hd), 1).contiguous(),
        }
        # ls * f(x) (+ x): without the shortcut the factor always folds into the last (activation-free) convolution; with it, in the
        # 16-bit modes (_fold_ls)
        fold = (not self.shortcut) or _fold_ls(dtype)
        pk["fusion_ls"] = _pack_bn_conv(self.fusion, dtype, device, self.ls_attn if fold else None)
        pk["ffn1_ls"] = _pack_bn_conv(self.ffn[1], dtype, device, self.ls_ffn if fold else None)
        pk["ls_attn"] = None if fold else _flat(self.ls_attn, device)
        pk["ls_ffn"] = None if fold else _flat(self.ls_ffn, device)
        return pk

    def _route(self, x, pk):
        B, H, W, _ = x.shape
        h = ops.conv2d(x, *pk["r0"], 1, 1, False, out_dtype=torch.float32)                 # [B,H,W,hp] fp32, pad channels = 0
        hn = torch.zeros((B, H, W, pk["r_hp"]), dtype=torch.float32, device=x.device)
        hid = pk["r_hid"]
        ops.group_norm(h[..., :hid], get_safe_groups(hid, 4), *pk["r1"], 1e-5, act="silu", out=hn[..., :hid])
        logits = ops.conv2d(hn, *pk["r3"], 1, 1, False)                                      # [B,H,W,4] fp32 (3 logits + pad)
        probs, _ = ops.token_softmax(logits, self.NUM_GROUPS, 1.0 / self.router.temperature)
        return probs

    def _head_tail(self, o, proj, norm):
        p = ops.conv2d(o, *proj, 1, 1, False)
        return ops.group_norm(p, get_safe_groups(self.dim, 8), *norm, 1e-5)

    def _run(self, x, out=None):
        B, H, W, C = x.shape
        pk = self._packed(x.device)
        lh, rh = self.local_head, self.region_head
        nh, scale = lh.num_heads, lh.head_dim ** -0.5      # the softmax scale is the TRUE head_dim's
        hd = pk["hdp"]                                      # kernels see heads padded to a multiple of 8 channels
        inner = nh * hd
        probs = self._route(x, pk)
        self.last_route = {"weights": probs}
        # sparse inference (moa/block.py:194-234): a batch-level decision — a head group whose gate stays at or below the threshold for
        # every token is not computed at all, the retained gates are renormalised per token.  One host sync, as in the reference.
        run, blend = (True, True, True), probs
        if self.sparse_inference:
            act, sparse_blend, mass = ops.moa_sparse_gate(probs, self.NUM_GROUPS, self.sparse_inference_threshold)
            self.last_route.update(active=act, executed_groups=sum(act), dropped_routing_mass=mass if not all(act) else 0.0)
            if not all(act):
                run, blend = tuple(act), sparse_blend
        heads = []
        if run[0]:   # local head (moa/heads.py:143-163): DW3x3 -> 1x1 qkv, v += DW7x7(v), 7x7-window attention
            qkv = ops.conv2d(ops.dwconv2d(x, pk["l_dw"], None, 3, False), *pk["l_qkv"], 1, 1, False)
            v = ops.dwconv2d(qkv[..., 2 * inner:], pk["l_pe"], None, 7, False, residual=qkv[..., 2 * inner:])
            win = max(1, min(lh.window_size, H, W))
            o = ops.window_attention(qkv[..., :inner], qkv[..., inner:2 * inner], v, nh, hd, scale, win)
            heads.append(self._head_tail(o, pk["l_proj"], pk["l_norm"]))
        if run[1]:   # regional head (moa/heads.py:208-253): full-resolution queries, pooled keys / values
            if min(H, W) <= 1:
                pooled = x
            else:
                stride = rh.pool_stride
                if rh.max_kv_tokens is not None:
                    while max(1, H // stride) * max(1, W // stride) > rh.max_kv_tokens:
                        stride *= 2
                pooled = ops.adaptive_avg_pool(x, max(1, H // stride), max(1, W // stride))
            kv = ops.conv2d(pooled, *pk["g_kv"], 1, 1, False)
            q = ops.conv2d(x, *pk["g_q"], 1, 1, False)
            o = ops.attention(q, kv[..., :inner], kv[..., inner:], nh, hd, scale)
            heads.append(self._head_tail(o, pk["g_proj"], pk["g_norm"]))
        if run[2]:   # global head (moa/heads.py:354-380): exact <= 512 tokens, blended with / replaced by random-feature attention
            qkv = ops.conv2d(x, *pk["a_qkv"], 1, 1, False)
            q, k, v = qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:]
            N = H * W
            if N <= self.LINEAR_ATTN_THRESHOLD:
                o = ops.attention(q, k, v, nh, hd, scale)
                start = self.LINEAR_ATTN_THRESHOLD - self.LINEAR_ATTN_BLEND_WINDOW
                if N > start:
                    o = ops.lerp(o, ops.linear_attention(q, k, v, pk["rf"], nh, hd), (N - start) / self.LINEAR_ATTN_BLEND_WINDOW)
            else:
                o = ops.linear_attention(q, k, v, pk["rf"], nh, hd)
            heads.append(self._head_tail(o, pk["a_proj"], pk["a_norm"]))
        mixed = ops.weighted_sum(blend, heads)       # (sparse: the active groups' gates sit in the first len(heads) columns)
        # x + ls_attn * fusion(mixed); then + ls_ffn * ffn(.) (moa/block.py:264-278); without the shortcut the same without the skips
        if not self.shortcut:
            x1 = ops.conv2d(mixed, *pk["fusion_ls"], 1, 1, False)
            return ops.conv2d(self.ffn[0]._run(x1), *pk["ffn1_ls"], 1, 1, False, out=out)
        x1 = _ls_conv(mixed, pk, "fusion_ls", "ls_attn", x)
        return _ls_conv(self.ffn[0]._run(x1), pk, "ffn1_ls", "ls_ffn", x1, out=out)


class C2fMoA(YmkModule):
    """moa/wrappers.py:40-142."""

    def __init__(self, c1, c2, n=1, num_heads=6, mlp_ratio=2.0, temperature=1.0, shortcut=True, e=0.5, aux_loss_coeff=0.01,
                 local_window_size=7, sequential_heads=True, regional_max_kv_tokens=4096, sparse_inference=False,
                 sparse_inference_threshold=0.02, inference_sparse_threshold=None):
        super().__init__()
        if inference_sparse_threshold is not None:    # moa/wrappers.py:82-89
            if sparse_inference_threshold != 0.02 and sparse_inference_threshold != inference_sparse_threshold:
                raise ValueError("Specify only one sparse inference threshold: sparse_inference_threshold or inference_sparse_threshold.")
            sparse_inference_threshold, sparse_inference = inference_sparse_threshold, True
        self.sparse_inference, self.sparse_inference_threshold = bool(sparse_inference), float(sparse_inference_threshold)
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        h, it = num_heads, 256
        while h % MoABlock.NUM_GROUPS != 0 and it > 0:
            h, it = h + 1, it - 1
        it = 256
        while self.c // h < 16 and h > MoABlock.NUM_GROUPS and it > 0:
            h, it = h - MoABlock.NUM_GROUPS, it - 1
        h = max(h, MoABlock.NUM_GROUPS)
        if h != num_heads:
            warnings.warn(f"C2fMoA(num_heads={num_heads}) adjusted to {h} (divisible by 3, head_dim >= 16)", stacklevel=2)
        self.m = nn.ModuleList(
            MoABlock(self.c, num_heads=h, mlp_ratio=mlp_ratio, temperature=temperature, shortcut=shortcut,
                     aux_loss_coeff=aux_loss_coeff, block_index=i, local_window_size=local_window_size,
                     sequential_heads=sequential_heads, regional_max_kv_tokens=regional_max_kv_tokens,
                     sparse_inference=sparse_inference, sparse_inference_threshold=sparse_inference_threshold) for i in range(n))

    def _run(self, x, out=None):
        """cv1 -> chunk(2) -> n MoABlocks chained on the last chunk -> cat -> cv2 (moa/wrappers.py:144-177); the
        concatenation is one buffer, every producer writes its channel slice."""
        B, H, W, _ = x.shape
        c, n = self.c, len(self.m)
        cat = ops.new_act(B, H, W, (2 + n) * c, x.dtype, x.device)
        self.cv1._run(x, out=cat[..., : 2 * c])
        for i, blk in enumerate(self.m):
            blk._run(cat[..., (1 + i) * c:(2 + i) * c], out=cat[..., (2 + i) * c:(3 + i) * c])
        return self.cv2._run(cat, out=out)


# ----------------------------------------------------------------------------------------- MoT
class _LocalConvTransformerExpert(nn.Module):
    """mot/experts.py:72-171."""

    def __init__(self, dim, num_heads, mlp_ratio=2.0, dropout=0.0, local_window_size=0):
        super().__init__()
        self.dim, self.num_heads, self.local_window_size = dim, num_heads, int(local_window_size)
        self.ls1 = nn.Parameter(torch.ones(dim, 1, 1) * 0.1)
        self.ls2 = nn.Parameter(torch.ones(dim, 1, 1) * 0.1)
        self.dw_mix = nn.Conv2d(dim, dim, 3, padding=1, groups=dim, bias=False)
        self.qkv = nn.Conv2d(dim, dim * 3, 1, bias=False)
        self.pe = nn.Conv2d(dim, dim, 7, padding=3, groups=dim, bias=False)
        self.proj = nn.Conv2d(dim, dim, 1, bias=False)
        self.norm1, self.norm2 = _gn(dim), _gn(dim)
        hid = int(dim * mlp_ratio)
        self.ffn_gate = nn.Sequential(Conv(dim, hid, 1), nn.Sigmoid())
        self.ffn_val = Conv(dim, hid, 1)
        self.ffn_out = Conv(hid, dim, 1, act=False)

    def pack(self, dtype, device):
        return {"n1": _pack_norm(self.norm1, device), "n2": _pack_norm(self.norm2, device), "dw": _pack_dw(self.dw_mix, dtype, device),
                "qkv": _pack_conv(self.qkv, dtype, device), "pe": _pack_dw(self.pe, dtype, device),
                "proj": _pack_conv(self.proj, dtype, device, scale=self.ls1 if _fold_ls(dtype) else None),
                "ffn_out": _pack_bn_conv(self.ffn_out, dtype, device, self.ls2 if _fold_ls(dtype) else None),
                "ffn_gv": _pack_bn_pair(self.ffn_gate[0], self.ffn_val, dtype, device),
                "ls1": None if _fold_ls(dtype) else _flat(self.ls1, device), "ls2": None if _fold_ls(dtype) else _flat(self.ls2, device)}

    def run(self, x, pk):
        """GN -> DW3x3 -> 1x1 qkv, v += DW7x7(v), attention (whole map, or local windows), proj, layer-scale residual;
        GN -> sigmoid(Conv) * Conv -> Conv, layer-scale residual (mot/experts.py:123-171)."""
        B, H, W, C = x.shape
        nh, hd = self.num_heads, C // self.num_heads
        g = get_safe_groups(C, 8)
        xn = ops.group_norm(x, g, *pk["n1"], 1e-5)
        qkv = ops.conv2d(ops.dwconv2d(xn, pk["dw"], None, 3, False), *pk["qkv"], 1, 1, False)
        v = ops.dwconv2d(qkv[..., 2 * C:], pk["pe"], None, 7, False, residual=qkv[..., 2 * C:])
        lws = self.local_window_size
        if lws > 0 and H * W > lws * lws:
            o = ops.window_attention(qkv[..., :C], qkv[..., C:2 * C], v, nh, hd, hd ** -0.5, lws)
        elif hd == 32 and (H * W <= 1024 or x.dtype == torch.float32):
            # whole-map attention with 32-wide heads is what the area-attention kernels of the A2C2f blocks compute: put v + pe(v) back into
            # the V slice of the [Q | K | V] buffer (stream-ordered after the stencil) and reuse them.  16-bit maps of more than 1024
            # tokens (the L-scale MoT blocks of BASELINE config 5: 6400 tokens at 1280 px) go to the streaming matrix-core kernel below,
            # which is what area_attn itself routes them to and takes q, k, v as separate views: no copy
            ops.copy_channels(v, qkv[..., 2 * C:])
            o = ops.area_attn(qkv, nh, 1)
        else:
            o = ops.attention(qkv[..., :C], qkv[..., C:2 * C], v, nh, hd, hd ** -0.5)
        x1 = _ls_conv(o, pk, "proj", "ls1", x)
        xn = ops.group_norm(x1, g, *pk["n2"], 1e-5)
        hid = pk["ffn_gv"][0].shape[0] // 2
        gv = ops.conv2d(xn, *pk["ffn_gv"], 1, 1, True)      # [SiLU(gate conv) | SiLU(value conv)]: two 1x1 convolutions of xn as one launch
        glu = ops.eltwise_mul(gv[..., :hid], gv[..., hid:], act_a="sigmoid")
        return _ls_conv(glu, pk, "ffn_out", "ls2", x1)


def _mlp(dim, hid, dropout):
    return nn.Sequential(nn.Linear(dim, hid), nn.GELU(), nn.Dropout(dropout), nn.Linear(hid, dim))


def _run_token_ffn(x1, ffn, norm2, ls2, pk):
    """LayerNorm -> Linear -> GELU -> Linear with layer scale (mot/experts.py:318-325, 486-493)."""
    h = ops.conv2d_act(ops.layer_norm(x1, *pk["n2"], 1e-5), *pk["f0"], 1, 1, "gelu")
    return _ls_conv(h, pk, "f3", "ls2", x1)


class _WindowTransformerExpert(nn.Module):
    """mot/experts.py:174-325."""

    def __init__(self, dim, num_heads, window_size=7, mlp_ratio=2.0, dropout=0.0, shift_size=0):
        super().__init__()
        self.num_heads, self.win = num_heads, window_size
        self.shift_size = (window_size // 2) if shift_size else 0
        self.ls1 = nn.Parameter(torch.ones(dim) * 0.1)
        self.ls2 = nn.Parameter(torch.ones(dim) * 0.1)
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim, bias=False)
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.ffn = _mlp(dim, int(dim * mlp_ratio), dropout)

    def pack(self, dtype, device):
        # tokens added by the window padding are zeros BEFORE the LayerNorm (experts.py:275-285): after it they equal the
        # LayerNorm bias, so their q / k / v are one constant vector each
        pad = (self.qkv.weight.detach().float() @ self.norm1.bias.detach().float()).to(device)
        C = self.norm1.bias.numel()
        return {"n1": _pack_norm(self.norm1, device), "n2": _pack_norm(self.norm2, device),
                "qkv": _pack_conv(self.qkv, dtype, device), "proj": _pack_conv(self.proj, dtype, device, scale=self.ls1 if _fold_ls(dtype) else None),
                "pad": tuple(pad[i * C:(i + 1) * C].contiguous() for i in range(3)),
                "f0": _pack_conv(self.ffn[0], dtype, device), "f3": _pack_conv(self.ffn[3], dtype, device, scale=self.ls2 if _fold_ls(dtype) else None),
                "ls1": None if _fold_ls(dtype) else _flat(self.ls1, device), "ls2": None if _fold_ls(dtype) else _flat(self.ls2, device)}

    def run(self, x, pk, qkv=None):
        C = x.shape[-1]
        nh, hd = self.num_heads, C // self.num_heads
        if qkv is None:     # (MoTBlock hands over its slice of the projection it shares with the deformable expert in the 16-bit modes)
            qkv = ops.conv2d(ops.layer_norm(x, *pk["n1"], 1e-5), *pk["qkv"], 1, 1, False)
        a = ops.window_attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], nh, hd, hd ** -0.5, self.win,
                                 shift=self.shift_size, pad_q=pk["pad"][0], pad_k=pk["pad"][1], pad_v=pk["pad"][2])
        x1 = _ls_conv(a, pk, "proj", "ls1", x)
        return _run_token_ffn(x1, self.ffn, self.norm2, self.ls2, pk)


class _DeformableTransformerExpert(nn.Module):
    """mot/experts.py:328-493."""

    def __init__(self, dim, num_heads, n_points=4, mlp_ratio=2.0, dropout=0.0, align_corners=True):
        super().__init__()
        self.num_heads, self.n_points, self.align_corners = num_heads, n_points, align_corners
        self.ls1 = nn.Parameter(torch.ones(dim) * 0.1)
        self.ls2 = nn.Parameter(torch.ones(dim) * 0.1)
        self.q_proj = nn.Linear(dim, dim, bias=False)
        self.v_proj = nn.Linear(dim, dim, bias=False)
        self.offset_proj = nn.Linear(dim, num_heads * n_points * 2, bias=True)
        self.attn_proj = nn.Linear(dim, num_heads * n_points, bias=True)
        self.out_proj = nn.Linear(dim, dim, bias=False)
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.ffn = _mlp(dim, int(dim * mlp_ratio), dropout)

    def pack(self, dtype, device):
        return {"n1": _pack_norm(self.norm1, device), "n2": _pack_norm(self.norm2, device),
                # q and v are two projections of the normalised tokens, offsets and attention logits two projections of q: one launch each pair
                "qv": _pack_conv_pair(self.q_proj, self.v_proj, dtype, device),
                "offaw": _pack_conv_pair(self.offset_proj, self.attn_proj, dtype, device,
                                         pad_cout_to=_ceil(self.offset_proj.out_features + self.attn_proj.out_features, 4)),
                "out": _pack_conv(self.out_proj, dtype, device, scale=self.ls1 if _fold_ls(dtype) else None),
                "f0": _pack_conv(self.ffn[0], dtype, device), "f3": _pack_conv(self.ffn[3], dtype, device, scale=self.ls2 if _fold_ls(dtype) else None),
                "ls1": None if _fold_ls(dtype) else _flat(self.ls1, device), "ls2": None if _fold_ls(dtype) else _flat(self.ls2, device)}

    def run(self, x, pk, qv=None):
        C = x.shape[-1]
        nh, hd, npnt = self.num_heads, C // self.num_heads, self.n_points
        if qv is None:
            qv = ops.conv2d(ops.layer_norm(x, *pk["n1"], 1e-5), *pk["qv"], 1, 1, False)      # [q | v]
        oa = ops.conv2d(qv[..., :C], *pk["offaw"], 1, 1, False, out_dtype=torch.float32)      # [offsets | attention logits], fp32 (sampling coordinates)
        off, aw = oa[..., : nh * npnt * 2], oa[..., nh * npnt * 2: nh * npnt * 3]
        o = ops.deform_attention(qv[..., C:], off, aw, nh, hd, npnt, self.align_corners)
        x1 = _ls_conv(o, pk, "out", "ls1", x)
        return _run_token_ffn(x1, self.ffn, self.norm2, self.ls2, pk)


class _MoTRouter(nn.Module):  # noqa: E302
    """mot/router.py:57-160: parameters of the token-level (1x1 -> GroupNorm -> SiLU -> 1x1) or image-level (GAP -> Linear -> SiLU ->
    Linear) router and of the optional scene-aware residual (Linear(3, h) -> SiLU -> Linear(h, E) on three statistics of the routed map,
    :145-160).  Same parameter names as the reference; the arithmetic is MoTBlock._route."""

    def __init__(self, dim, num_experts=3, top_k=2, temperature=1.0, use_spatial=True, scene_aware=False, scene_hidden_dim=None,
                 scene_inference_mode="dynamic"):
        super().__init__()
        self.num_experts, self.top_k, self.use_spatial = num_experts, top_k, bool(use_spatial)
        self.register_buffer("temperature", torch.tensor(max(temperature, 0.1)), persistent=True)
        hidden = max(dim // 8, num_experts * 4)
        if use_spatial:
            self.router = nn.Sequential(nn.Conv2d(dim, hidden, 1, bias=False), _gn(hidden, 4), nn.SiLU(),
                                        nn.Conv2d(hidden, num_experts, 1, bias=True))
        else:
            self.router = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(dim, hidden, bias=False), nn.SiLU(),
                                        nn.Linear(hidden, num_experts, bias=True))
        nn.init.zeros_(self.router[-1].weight)
        nn.init.zeros_(self.router[-1].bias)
        self.scene_aware = False
        self.scene_hidden_dim = scene_hidden_dim
        self.scene_projector = None
        self.last_scene_stats = None
        self._last_scene_bias = None
        self.last_scene_applied = False
        self.last_scene_bypass_reason = None
        self.set_scene_inference_mode(scene_inference_mode)
        if scene_aware:
            self.enable_scene_aware(scene_hidden_dim)

    @property
    def last_scene_bias(self):
        """The scene projector's per-image logit residual of the last forward (mot/router.py:224-240); None when not applied."""
        v = self._last_scene_bias
        if callable(v):
            v = self._last_scene_bias = v()
        return v

    @last_scene_bias.setter
    def last_scene_bias(self, v):
        self._last_scene_bias = v

    def set_scene_inference_mode(self, mode):
        """mot/router.py:138-143."""
        normalized = str(mode).strip().lower()
        if normalized not in {"dynamic", "bypass"}:
            raise ValueError("scene_inference_mode must be 'dynamic' or 'bypass'")
        self.scene_inference_mode = normalized

    def enable_scene_aware(self, hidden_dim=None):
        """mot/router.py:145-160 (zero-initialised residual; parameters created once)."""
        if self.scene_projector is None:
            hidden = int(hidden_dim or self.scene_hidden_dim or 3)
            if hidden <= 0:
                raise ValueError("scene_hidden_dim must be positive")
            self.scene_projector = nn.Sequential(nn.Linear(3, hidden), nn.SiLU(), nn.Linear(hidden, self.num_experts))
            nn.init.zeros_(self.scene_projector[-1].weight)
            nn.init.zeros_(self.scene_projector[-1].bias)
            self.scene_hidden_dim = hidden
        self.scene_aware = True


class MoTBlock(YmkModule):
    """mot/block.py:20-170."""

    NUM_EXPERTS = 3

    def __init__(self, dim, num_heads=8, top_k=2, window_size=7, n_points=4, mlp_ratio=2.0, temperature=1.0,
                 use_spatial_router=True, balance_loss_coeff=0.01, router_z_loss_coeff=None, dropout=0.0,
                 exploration_eps=0.02, window_shift=False, grid_align_corners=True, sparse_train=False,
                 scene_aware_router=False, scene_hidden_dim=None, scene_consistency_coeff=0.0, sparse_train_warmup_steps=0,
                 scene_inference_mode="dynamic", local_attn_window=0):
        super().__init__()
        if not 1 <= top_k <= self.NUM_EXPERTS:
            raise ValueError(f"top_k must be in [1, {self.NUM_EXPERTS}], got {top_k}")
        self.top_k = int(top_k)
        self.register_buffer("_sparse_train_step", torch.tensor(0, dtype=torch.long), persistent=True)
        h = num_heads
        while dim % h != 0 and h > 1:
            h -= 1
        h = max(1, h)
        self.experts = nn.ModuleList([
            _LocalConvTransformerExpert(dim, h, mlp_ratio, dropout, local_window_size=local_attn_window),
            _WindowTransformerExpert(dim, h, window_size, mlp_ratio, dropout, shift_size=window_size // 2 if window_shift else 0),
            _DeformableTransformerExpert(dim, h, n_points, mlp_ratio, dropout, align_corners=grid_align_corners)])
        self.router = _MoTRouter(dim, self.NUM_EXPERTS, top_k, temperature=temperature, use_spatial=use_spatial_router,
                                 scene_aware=scene_aware_router, scene_hidden_dim=scene_hidden_dim, scene_inference_mode=scene_inference_mode)
        self.out_norm = _gn(dim)
        self.out_proj = nn.Conv2d(dim, dim, 1, bias=False)

    def _pack(self, dtype, device):
        r = self.router.router
        first = r[0] if self.router.use_spatial else r[2]
        hid = first.out_channels if self.router.use_spatial else first.out_features
        hp = _ceil(hid, 4)
        dim, nh = (first.in_channels if self.router.use_spatial else first.in_features), self.experts[0].num_heads
        if (dim // nh) % 8:
            raise NotImplementedError(f"ymk MoTBlock: head_dim {dim // nh} is not a multiple of 8 (C2fMoT keeps head_dim >= 8 and "
                                      "dividing the width; widths that are multiples of 64 always qualify)")
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()   # noqa: E731
        sp = self.router.scene_projector
        scene = None if sp is None else {"w1": f32(sp[0].weight), "b1": f32(sp[0].bias), "w2": f32(sp[2].weight), "b2": f32(sp[2].bias)}
        if self.router.use_spatial:
            head = {"r0": _pack_conv(r[0], dtype, device, pad_cout_to=hp), "r1": _pack_norm(r[1], device), "r_hid": hid, "r_hp": hp,
                    "r3": _pack_conv(r[3], torch.float32, device, pad_cout_to=4, pad_cin_to=hp)}
        else:   # image-level router: Linear(dim, hidden, bias=False) -> SiLU -> Linear(hidden, E) on the pooled map, as fp32 1x1 convolutions
            head = {"g0": _pack_conv(r[2], torch.float32, device, pad_cout_to=hp),
                    "g1": _pack_conv(r[4], torch.float32, device, pad_cout_to=4, pad_cin_to=hp)}
        return {**head, "scene": scene,
                "inv_temp": 1.0 / float(self.router.temperature),
                "experts": [e.pack(dtype, device) for e in self.experts],
                "shared": self._pack_shared(dtype, device) if _fold_ls(dtype) else None,
                "out_proj": _pack_conv(self.out_proj, dtype, device), "out_norm": _pack_norm(self.out_norm, device)}

    def _pack_shared(self, dtype, device):
        """The window and the deformable expert both start with LayerNorm(x) -> bias-free projections; the two LayerNorms differ only in
        their affine parameters.  16-bit modes: ONE affine-free LayerNorm, and ONE convolution [window q | k | v | deformable q | v] with each
        LayerNorm's weight folded into its projection's columns and its bias into the convolution's bias (W (xhat * g + b) = (W diag g) xhat
        + W b): one normalisation pass and one launch less per block.  fp32 keeps the reference's order of operations (see _fold_ls)."""
        import types

        win, dfm = self.experts[1], self.experts[2]
        rows, biases = [], []
        for lin, ln in ((win.qkv, win.norm1), (dfm.q_proj, dfm.norm1), (dfm.v_proj, dfm.norm1)):
            if lin.bias is not None:
                return None
            w = lin.weight.detach().float()
            rows.append(w * ln.weight.detach().float().view(1, -1))
            biases.append(w @ ln.bias.detach().float())
        both = types.SimpleNamespace(weight=torch.cat(rows, 0), bias=torch.cat(biases, 0), groups=1)
        C = win.norm1.weight.numel()
        return {"w": _pack_conv(both, dtype, device), "ones": torch.ones(C, device=device), "zeros": torch.zeros(C, device=device),
                "eps": float(win.norm1.eps)}

    def _route(self, x, pk):
        """_MoTRouter.forward, eval (mot/router.py:224-295): per-token weights fp32 [B,H,W,E] (zeros outside the top-k) and the per-image
        "expert e is used" flags.  Token-level logits come from the 1x1 -> GroupNorm -> SiLU -> 1x1 head; the image-level router's logits
        and the scene-aware residual are per-image rows added inside the softmax kernel (ymk_token_softmax `bias`)."""
        B, H, W, C = x.shape
        rt = self.router
        logits = base = None
        if rt.use_spatial:
            h = ops.conv2d(x, *pk["r0"], 1, 1, False, out_dtype=torch.float32)
            hn = torch.zeros((B, H, W, pk["r_hp"]), dtype=torch.float32, device=x.device)
            hid = pk["r_hid"]
            ops.group_norm(h[..., :hid], get_safe_groups(hid, 4), *pk["r1"], 1e-5, act="silu", out=hn[..., :hid])
            logits = ops.conv2d(hn, *pk["r3"], 1, 1, False)
        else:
            pooled = ops.channel_stats(x)                                           # AdaptiveAvgPool2d(1) in fp32: [B,1,1,C]
            hmid = ops.conv2d(pooled, *pk["g0"], 1, 1, True)                         # Linear -> SiLU
            base = ops.conv2d(hmid, *pk["g1"], 1, 1, False).reshape(B, -1)[:, : self.NUM_EXPERTS].contiguous()
        apply_scene = bool(rt.scene_aware and rt.scene_inference_mode == "dynamic")   # (eval: mot/router.py:224-229)
        rt.last_scene_applied = apply_scene
        rt.last_scene_bypass_reason = "inference_policy_bypass" if rt.scene_aware and not apply_scene else None
        rt.last_scene_stats = rt.last_scene_bias = None
        bias = base
        if apply_scene:
            sc = pk["scene"]
            if sc is None:
                raise RuntimeError("scene-aware MoT router is enabled without a scene projector")
            stats, bias = ops.scene_bias(x, sc["w1"], sc["b1"], sc["w2"], sc["b2"], base=base)
            rt.last_scene_stats = stats
            # mot/router.py:224-240 publishes the PURE scene bias.  With an image-level router the kernel returns base logits + bias in one
            # pass (what the softmax consumes); the diagnostic is then the second call's output without the base — off the hot path: only
            # when somebody reads it (`_MoTRouter.last_scene_bias` resolves the thunk)
            rt.last_scene_bias = bias if base is None else (lambda: ops.scene_bias(x, sc["w1"], sc["b1"], sc["w2"], sc["b2"], base=None)[1])
        return ops.token_softmax(logits, self.NUM_EXPERTS, pk["inv_temp"], top_k=self.top_k, bias=bias, shape=(B, H, W))

    def _run(self, x, out=None):
        """mot/block.py:298-417, eval.  The reference runs expert e only on the images where some token selected it;
        every expert is image-local, so running all of them on the whole batch and weighting per token (weight 0 where
        the expert was not selected) gives the same result without a device->host read of the routing decision."""
        B, H, W, C = x.shape
        pk = self._packed(x.device)
        weights, active = self._route(x, pk)
        self.last_route = {"weights": weights, "active": active}
        sh = pk["shared"]
        if sh is not None and float(self.experts[2].norm1.eps) == sh["eps"]:
            xhat = ops.layer_norm(x, sh["ones"], sh["zeros"], sh["eps"])
            proj = ops.conv2d(xhat, *sh["w"], 1, 1, False)                     # [window q | k | v | deformable q | v]
            outs = [self.experts[0].run(x, pk["experts"][0]), self.experts[1].run(x, pk["experts"][1], qkv=proj[..., : 3 * C]),
                    self.experts[2].run(x, pk["experts"][2], qv=proj[..., 3 * C:])]
        else:
            outs = [e.run(x, p) for e, p in zip(self.experts, pk["experts"])]
        mixed = ops.weighted_sum(weights, outs)
        p = ops.conv2d(mixed, *pk["out_proj"], 1, 1, False)
        return ops.group_norm(p, get_safe_groups(C, 8), *pk["out_norm"], 1e-5, residual=x, out=out)


class C2fMoT(YmkModule):
    """mot/wrappers.py:19-105."""

    def __init__(self, c1, c2, n=1, num_heads=6, top_k=2, window_size=7, n_points=4, mlp_ratio=2.0, temperature=1.0,
                 balance_loss_coeff=0.01, e=0.5, sparse_train=False, scene_aware_router=False, scene_hidden_dim=None,
                 scene_consistency_coeff=0.0, sparse_train_warmup_steps=0, scene_inference_mode="dynamic", local_attn_window=0):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        h = num_heads
        while h > 1 and (self.c % h != 0 or self.c // h < 8):
            h -= 1
        h = max(1, h)
        self.m = nn.ModuleList(
            MoTBlock(dim=self.c, num_heads=h, top_k=top_k, window_size=window_size, n_points=n_points, mlp_ratio=mlp_ratio,
                     temperature=temperature, balance_loss_coeff=balance_loss_coeff, window_shift=bool(i % 2),
                     sparse_train=sparse_train, scene_aware_router=scene_aware_router, scene_hidden_dim=scene_hidden_dim,
                     scene_consistency_coeff=scene_consistency_coeff, sparse_train_warmup_steps=sparse_train_warmup_steps,
                     scene_inference_mode=scene_inference_mode, local_attn_window=local_attn_window) for i in range(n))

    def _run(self, x, out=None):
        """cv1 -> chunk(2) -> n MoTBlocks chained on the last chunk (odd blocks shifted) -> cat -> cv2 (mot/wrappers.py:107-114)."""
        B, H, W, _ = x.shape
        c, n = self.c, len(self.m)
        cat = ops.new_act(B, H, W, (2 + n) * c, x.dtype, x.device)
        self.cv1._run(x, out=cat[..., : 2 * c])
        for i, blk in enumerate(self.m):
            blk._run(cat[..., (1 + i) * c:(2 + i) * c], out=cat[..., (2 + i) * c:(3 + i) * c])
        return self.cv2._run(cat, out=out)


GATED_CHAIN = (AdaptiveGateMoE, FusedAdaptiveGateMoE, HybridAdaptiveGateMoE, HybridAdaptiveGateMoEv2, LowRankHybridAdaptiveGateMoE,
               RefinedLowRankHybridAdaptiveGateMoE, DetailAwareLowRankHybridAdaptiveGateMoE, ContextRefinedLowRankHybridAdaptiveGateMoE,
               VisualEnhancedAdaptiveGateMoE)    # YAML generations v0_4 ... v0_11 (one class per generation)
# ----------------------------------------------------------------------------------------- v0_1: ModularRouterExpertMoE
class EfficientSpatialRouter(nn.Module):
    """Parameter container with the reference's names (moe/routers.py:268-281): router = Conv3x3 -> BN -> SiLU -> Conv1x1 -> BN."""

    def __init__(self, in_channels, num_experts, reduction=8, top_k=2, noise_std=1.0, pool_scale=4):
        super().__init__()
        self.num_experts, self.top_k, self.noise_std, self.pool_scale = num_experts, top_k, noise_std, pool_scale
        red = max(in_channels // reduction, 8)
        self.router = nn.Sequential(nn.Conv2d(in_channels, red, 3, padding=1, bias=False), nn.BatchNorm2d(red), nn.SiLU(inplace=False),
                                    nn.Conv2d(red, num_experts, 1, bias=False), nn.BatchNorm2d(num_experts))


class SimpleExpert(nn.Module):
    """moe/experts.py:73-88: conv = 1x1 -> GN -> SiLU -> 1x1 -> GN."""

    def __init__(self, in_channels, out_channels, expand_ratio=2, num_groups=8):
        super().__init__()
        hid = int(in_channels * expand_ratio)
        self.conv = nn.Sequential(nn.Conv2d(in_channels, hid, 1, bias=False), _gn(hid, num_groups), nn.SiLU(inplace=True),
                                  nn.Conv2d(hid, out_channels, 1, bias=False), _gn(out_channels, num_groups))


class ModularRouterExpertMoE(YmkModule):
    """The MoE block of the v0_1 master YAMLs (= `OptimizedMOEImproved`, moe/modules.py:957-1198, alias :1744) in the configuration
    those YAMLs use (`[c2, num_experts, top_k]`: EfficientSpatialRouter, SimpleExpert, shared expert, residual).  Eval forward on
    libymk, true sparse dispatch:

        router   4x4 average pool -> Conv3x3+BN+SiLU -> Conv1x1+BN (fp32 out) -> spatial mean -> softmax / top-k / renormalise
                 (`ymk_gated_route_decide` with the global stream and the complexity gate switched off)
        experts  only the routed filter banks run (`ymk_expert_conv_glds`): 1x1 -> GroupNorm(affine row of the image's expert) -> SiLU
                 -> 1x1 -> GroupNorm, slot-major [k * B] maps
        output   sum_j w_j expert_j(x) + SiLU(BN(1x1 x)) + x in one weighted sum
    Other router / expert types of the reference constructor are not on the YAML surface and raise."""

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, expert_type="simple", router_type="efficient", noise_std=1.0,
                 balance_loss_coeff=1.0, router_z_loss_coeff=1.0, expert_expand_ratio=2.0, progressive_sparsity=True, detach_routing=False,
                 add_residual=True):
        super().__init__()
        if expert_type != "simple" or router_type != "efficient":
            raise NotImplementedError("ymk ModularRouterExpertMoE: the YAML surface uses the default 'efficient' router and 'simple' experts")
        if not 1 <= top_k <= 3 or top_k > num_experts:
            raise ValueError("ymk ModularRouterExpertMoE: 1 <= top_k <= min(3, num_experts)")
        self.in_channels, self.out_channels, self.num_experts, self.top_k = in_channels, out_channels, num_experts, top_k
        self.balance_loss_coeff, self.router_z_loss_coeff, self.add_residual = balance_loss_coeff, router_z_loss_coeff, add_residual
        self.routing = EfficientSpatialRouter(in_channels, num_experts, top_k=top_k, noise_std=noise_std)
        self.experts = nn.ModuleList(SimpleExpert(in_channels, out_channels, expand_ratio=expert_expand_ratio) for _ in range(num_experts))
        self.shared_expert = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels), nn.SiLU(inplace=True))
        self.last_route = {}

    @staticmethod
    def _fold(conv, bn, device):
        return ops.fold_bn(conv.weight.detach().float().to(device), bn.weight.float().to(device), bn.bias.float().to(device),
                           bn.running_mean.float().to(device), bn.running_var.float().to(device), bn.eps)

    def _pack(self, dtype, device):
        f32 = torch.float32
        E, r = self.num_experts, self.routing.router
        E4 = _ceil(E, 4)
        w0, b0 = self._fold(r[0], r[1], device)
        w3, b3 = self._fold(r[3], r[4], device)
        red, rp = w0.shape[0], _ceil(w0.shape[0], 8)
        w0 = torch.cat([w0, w0.new_zeros((rp - red, *w0.shape[1:]))], 0)
        b0 = torch.cat([b0, b0.new_zeros(rp - red)], 0)
        w3 = torch.cat([w3, w3.new_zeros((w3.shape[0], rp - red, 1, 1))], 1)           # zero columns for the padded hidden channels (SiLU(0) = 0)
        w3 = torch.cat([w3, w3.new_zeros((E4 - E, *w3.shape[1:]))], 0)
        b3 = torch.cat([b3, b3.new_zeros(E4 - E)], 0)
        ws, bs = self._fold(self.shared_expert[0], self.shared_expert[1], device)
        return {
            "r0": (ops.pack_conv_weight(w0, dtype), b0.contiguous()), "r3": (ops.pack_conv_weight(w3, dtype), b3.contiguous()),
            "sh": (ops.pack_conv_weight(ws, dtype), bs.contiguous()),
            "e1": torch.stack([_pack_conv(e.conv[0], dtype, device)[0] for e in self.experts]).contiguous(),          # [E][hid][Kpad]
            "n1": (torch.stack([e.conv[1].weight.detach().float() for e in self.experts]).to(device).contiguous(),
                   torch.stack([e.conv[1].bias.detach().float() for e in self.experts]).to(device).contiguous()),
            "e2": torch.stack([_pack_conv(e.conv[3], dtype, device)[0] for e in self.experts]).contiguous(),          # [E][cout][Kpad]
            "n2": (torch.stack([e.conv[4].weight.detach().float() for e in self.experts]).to(device).contiguous(),
                   torch.stack([e.conv[4].bias.detach().float() for e in self.experts]).to(device).contiguous()),
            "consts": {},   # per batch size: the (absent) global stream of the decision kernel (zeros) and a complexity logit of +100
        }                   # (sigmoid = 1: every ranked expert kept), allocated once

    def _run(self, x, out=None):
        B, H, W, C = x.shape
        pk = self._packed(x.device)
        E, k, gs = self.num_experts, self.top_k, get_safe_groups
        ps = self.routing.pool_scale
        xin = ops.avg_pool(x, ps) if (H > ps and W > ps) else x
        h = ops.conv2d(xin, *pk["r0"], 3, 1, True)
        logits = ops.channel_stats(ops.conv2d(h, *pk["r3"], 1, 1, False, out_dtype=torch.float32))[..., :E]       # spatial mean in fp32 (routers.py:300)
        if B not in pk["consts"]:
            pk["consts"][B] = (torch.zeros((B, 1, 1, _ceil(E, 4)), dtype=torch.float32, device=x.device),
                               torch.full((B, 1, 1, 1), 100.0, dtype=torch.float32, device=x.device))
        g0, cp = pk["consts"][B]
        w, idx, probs, rows = ops.gated_route_decide(g0[..., :E], logits, -100.0, 1.0, k, cp, clamp=0)   # `_process_logits`: plain softmax (routers.py:207)
        self.last_route = {"weights": w, "indices": idx, "probs": probs}
        hid, cout = pk["e1"].shape[1], self.out_channels
        f = ops.expert_conv(x, pk["e1"], 1, idx)                                                                   # [k * B, H, W, hid], slot-major
        f = ops.group_norm(f, gs(hid, 8), *pk["n1"], 1e-5, act="silu", affine_rows=rows)
        f = ops.expert_conv(f, pk["e2"], 1, rows.view(-1, 1).contiguous())                                         # image n of the slot-major batch -> its own expert
        f = ops.group_norm(f, gs(cout, 8), *pk["n2"], 1e-5, affine_rows=rows)
        res = x if (self.add_residual and C == cout) else None
        if res is not None and res.stride(2) != C:                                                                 # a dense copy of a channel-slice view
            res = ops.copy_channels(res, torch.empty((B, H, W, C), dtype=x.dtype, device=x.device))
        shared = ops.conv2d(x, *pk["sh"], 1, 1, True, residual=res)                                                # SiLU(BN(1x1 x)) + x
        wk = torch.ones((B, 1, 1, 4), dtype=torch.float32, device=x.device)
        wk[..., :k].copy_(w)
        return ops.weighted_sum(wk, [f[j * B:(j + 1) * B] for j in range(k)] + [shared], out=out)


# ----------------------------------------------------------------------------------------- v0_1 uomoe / exp v0_2: UltraOptimizedMoE
class UltraEfficientRouter(nn.Module):
    """Parameter container with the reference's names (moe/routers.py:69-95): router = DW3x3 -> GN(8) -> SiLU -> 1x1 -> GN(4) -> SiLU -> 1x1 (+bias)."""

    def __init__(self, in_channels, num_experts, reduction=16, top_k=2, noise_std=1.0, temperature=1.0, pool_scale=8):
        super().__init__()
        self.num_experts, self.top_k, self.noise_std, self.pool_scale = num_experts, top_k, noise_std, pool_scale
        self.temperature = max(float(temperature), 1e-3)
        red = max(in_channels // reduction, 4)
        self.router = nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, padding=1, groups=in_channels, bias=False), _gn(in_channels, 8),
                                    nn.SiLU(inplace=False), nn.Conv2d(in_channels, red, 1, bias=False), _gn(red, 4), nn.SiLU(inplace=False),
                                    nn.Conv2d(red, num_experts, 1, bias=True))
        self.softmax = nn.Softmax(dim=1)


class UltraOptimizedMoE(YmkModule):
    """The MoE block of `v0_1/det/yolo-master-n-uomoe*.yaml` and `exp/yolo-master-v0_2.yaml` (moe/modules.py:121-232; rows `[c2, num_experts,
    top_k]`: UltraEfficientRouter, OptimizedSimpleExpert — the same 1x1 -> GN -> SiLU -> 1x1 -> GN body as v0_1's SimpleExpert —, the
    always-on shared expert, no residual).  Eval forward on libymk, true sparse dispatch:

        router   8x8 average pool (fp32) -> DW3x3 -> GN -> SiLU -> 1x1 -> GN -> SiLU -> 1x1 + bias -> per-pixel softmax of the clamped logits,
                 mean over the pixels, top-k, renormalise, weights <= 0.01 dropped (`ymk_pooled_softmax_route`; moe/routers.py:97-147,
                 moe/utils.py:166-169)
        experts  only the routed filter banks run (`ymk_expert_conv_glds`), GroupNorm with the routed expert's affine row, slot-major maps
        output   SiLU(GN(1x1 x)) + clamp(sum_j w_j expert_j(x), +-1e4)      (moe/utils.py:181-203, moe/modules.py:224)
    Other expert types of the reference constructor ("ghost", "inverted") are not on the YAML surface and raise."""

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, expert_type="simple", router_reduction=16, router_pool_scale=8,
                 noise_std=1.0, router_temperature=1.0, balance_loss_coeff=1.0, router_z_loss_coeff=1.0, num_groups=8, weight_threshold=0.01):
        super().__init__()
        if expert_type != "simple":
            raise NotImplementedError("ymk UltraOptimizedMoE: the YAML surface uses the default 'simple' experts")
        if not 1 <= top_k <= 3 or top_k > num_experts or num_experts > 32:
            raise ValueError("ymk UltraOptimizedMoE: 1 <= top_k <= min(3, num_experts), num_experts <= 32")
        self.in_channels, self.out_channels, self.num_experts, self.top_k, self.expert_type = in_channels, out_channels, num_experts, top_k, expert_type
        self.balance_loss_coeff, self.router_z_loss_coeff, self.weight_threshold, self.num_groups = balance_loss_coeff, router_z_loss_coeff, weight_threshold, num_groups
        self.routing = UltraEfficientRouter(in_channels, num_experts, reduction=router_reduction, top_k=top_k, noise_std=noise_std,
                                            temperature=router_temperature, pool_scale=router_pool_scale)
        self.experts = nn.ModuleList(SimpleExpert(in_channels, out_channels, expand_ratio=2, num_groups=num_groups) for _ in range(num_experts))
        self.shared_expert = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, bias=False), _gn(out_channels, num_groups), nn.SiLU(inplace=True))
        self.last_route = {}

    def _pack(self, dtype, device):
        f32 = torch.float32
        E, r = self.num_experts, self.routing.router
        red = r[3].out_channels
        rp, E4 = _ceil(red, 4), _ceil(E, 4)
        return {
            "r0": _pack_dw(r[0], f32, device), "r1": _pack_norm(r[1], device),
            "r3": _pack_conv(r[3], f32, device, pad_cout_to=rp), "r4": _pack_norm(r[4], device), "red": red, "rp": rp,
            "r6": _pack_conv(r[6], f32, device, pad_cout_to=E4, pad_cin_to=rp),
            "sh": _pack_conv(self.shared_expert[0], dtype, device), "shn": _pack_norm(self.shared_expert[1], device),
            "e1": torch.stack([_pack_conv(e.conv[0], dtype, device)[0] for e in self.experts]).contiguous(),          # [E][hid][Kpad]
            "n1": (torch.stack([e.conv[1].weight.detach().float() for e in self.experts]).to(device).contiguous(),
                   torch.stack([e.conv[1].bias.detach().float() for e in self.experts]).to(device).contiguous()),
            "e2": torch.stack([_pack_conv(e.conv[3], dtype, device)[0] for e in self.experts]).contiguous(),          # [E][cout][Kpad]
            "n2": (torch.stack([e.conv[4].weight.detach().float() for e in self.experts]).to(device).contiguous(),
                   torch.stack([e.conv[4].bias.detach().float() for e in self.experts]).to(device).contiguous()),
        }

    def _run(self, x, out=None):
        """UltraOptimizedMoE.forward, eval (moe/modules.py:212-232)."""
        B, H, W, C = x.shape
        pk = self._packed(x.device)
        E, k, gs, ng = self.num_experts, self.top_k, get_safe_groups, self.num_groups
        ps = self.routing.pool_scale
        xin = ops.avg_pool(x, ps if (H > ps and W > ps) else 1, out_dtype=torch.float32)          # the router runs in fp32
        h = ops.group_norm(ops.dwconv2d(xin, pk["r0"], None, 3, False), gs(C, 8), *pk["r1"], 1e-5, act="silu")
        h = ops.conv2d(h, *pk["r3"], 1, 1, False)
        red, rp = pk["red"], pk["rp"]
        hn = h if red == rp else torch.zeros(h.shape, dtype=h.dtype, device=h.device)             # pad channels must stay zero
        ops.group_norm(h[..., :red], gs(red, 4), *pk["r4"], 1e-5, act="silu", out=hn[..., :red])
        logits = ops.conv2d(hn, *pk["r6"], 1, 1, False)
        w, idx, pooled, rows = ops.pooled_softmax_route(logits, E, 1.0 / self.routing.temperature, k, 0.01)   # eval threshold: the constant of moe/utils.py:172 (the module attribute is not consulted there)
        self.last_route = {"weights": w, "indices": idx, "probs": pooled}
        hid, cout = pk["e1"].shape[1], self.out_channels
        f = ops.expert_conv(x, pk["e1"], 1, idx)                                                                   # [k * B, H, W, hid], slot-major
        f = ops.group_norm(f, gs(hid, ng), *pk["n1"], 1e-5, act="silu", affine_rows=rows)
        f = ops.expert_conv(f, pk["e2"], 1, rows.view(-1, 1).contiguous())                                         # image n of the slot-major batch -> its own expert
        f = ops.group_norm(f, gs(cout, ng), *pk["n2"], 1e-5, affine_rows=rows)
        wk = torch.zeros((B, 1, 1, 4), dtype=torch.float32, device=x.device)
        wk[..., :k].copy_(w)
        mix = ops.weighted_sum(wk, [f[j * B:(j + 1) * B] for j in range(k)])
        shared = ops.group_norm(ops.conv2d(x, *pk["sh"], 1, 1, False), gs(cout, ng), *pk["shn"], 1e-5, act="silu")
        return ops.clamp_add(mix, shared, 1e4, out=out)


# ----------------------------------------------------------------------------------------- v0_3: UltimateOptimizedMoE
class ZeroCostRouter(nn.Module):
    """Parameter container with the reference's names (moe/gated.py:938-961): router = Sequential(Linear(2C -> E, no bias), Softmax)."""

    def __init__(self, in_channels, num_experts, top_k, temperature=1.0):
        super().__init__()
        self.num_experts, self.top_k, self.temperature = num_experts, top_k, temperature
        self.router = nn.Sequential(nn.Linear(2 * in_channels, num_experts, bias=False), nn.Softmax(dim=1))


class _BalanceControllerState(nn.Module):
    """AdaptiveBalanceController (moe/gated.py:1767-1805) as far as inference goes: the `expert_importance` entry of the state_dict."""

    def __init__(self, num_experts):
        super().__init__()
        self.expert_importance = nn.Parameter(torch.ones(num_experts))


class UltimateOptimizedMoE(YmkModule):
    """The MoE block of the v0_3 master YAMLs (moe/modules.py:1534-1700; rows `[c2, num_experts, top_k, split_ratio]`).  Eval forward on
    libymk: channel split; static DW3x3+BN+SiLU -> 1x1+BN+SiLU; ZeroCostRouter (gated.py:963-992: Linear over the [mean | std] channel
    statistics, the router's OWN softmax, / temperature, clamp, a second softmax, top-k — two passes of `ymk_gated_route_decide`, the
    first for its probabilities); batch-level complexity scale on the routing weights (`ymk_batch_scale`); FusedExpertGroup with true
    sparse dispatch (only the routed rows of the grouped 3x3 run) + affine-free GroupNorm with the routed expert's affine row + SiLU;
    [static | dynamic] -> 1x1 -> GroupNorm + x."""

    def __init__(self, in_channels, out_channels, num_experts=4, top_k=2, split_ratio=0.5, num_groups=8, use_routing_cache=True,
                 capacity_factor=1.5, initial_temperature=2.0, final_temperature=0.5, entropy_coeff=0.01):
        super().__init__()
        self.in_channels, self.out_channels, self.num_experts, self.top_k, self.num_groups = in_channels, out_channels, num_experts, top_k, num_groups
        self.capacity_factor, self.initial_temperature, self.final_temperature, self.entropy_coeff = capacity_factor, initial_temperature, final_temperature, entropy_coeff
        self.dynamic_channels = int(in_channels * split_ratio)
        self.static_channels = in_channels - self.dynamic_channels
        self.out_dynamic = int(out_channels * split_ratio)
        self.out_static = out_channels - self.out_dynamic
        sc = self.static_channels
        self.static_net = nn.Sequential(
            nn.Conv2d(sc, sc, 3, padding=1, groups=sc, bias=False), nn.BatchNorm2d(sc), nn.SiLU(inplace=True),
            nn.Conv2d(sc, self.out_static, 1, bias=False), nn.BatchNorm2d(self.out_static), nn.SiLU(inplace=True))
        self.routing = ZeroCostRouter(self.dynamic_channels, num_experts, top_k, temperature=initial_temperature)
        self.fused_experts = FusedExpertGroup(self.dynamic_channels, self.out_dynamic, num_experts, num_groups)
        self.complexity_estimator = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(self.dynamic_channels, 1, 1), nn.Sigmoid())
        self.register_buffer("training_step", torch.tensor(0), persistent=False)
        self.register_buffer("current_top_k", torch.tensor(num_experts))
        self.balance_loss_coeff, self.router_z_loss_coeff = 1.0, 0.0
        self.balance_controller = _BalanceControllerState(num_experts)     # training-only; its buffer is part of the checkpoint contract
        self.proj = nn.Conv2d(out_channels, out_channels, 1, bias=False)
        self.bn = _gn(out_channels, num_groups)
        self.last_route = {}

    def _pack(self, dtype, device):
        f32 = torch.float32
        dyn, E = self.dynamic_channels, self.num_experts
        if dyn % 8 or self.static_channels % 8 or self.out_dynamic % 8 or self.out_static % 8:
            raise NotImplementedError(f"ymk UltimateOptimizedMoE: split {self.static_channels}+{dyn} breaks the 16-byte channel-vector rule")
        sn = self.static_net
        dw_w, dw_b = ops.fold_bn(sn[0].weight.detach().float().to(device), sn[1].weight.float().to(device), sn[1].bias.float().to(device),
                                 sn[1].running_mean.float().to(device), sn[1].running_var.float().to(device), sn[1].eps)
        pw_w, pw_b = ops.fold_bn(sn[3].weight.detach().float().to(device), sn[4].weight.float().to(device), sn[4].bias.float().to(device),
                                 sn[4].running_mean.float().to(device), sn[4].running_var.float().to(device), sn[4].eps)
        fe = self.fused_experts
        fc = fe.fused_conv
        w = fc.weight.detach().float().to(device)                       # [E*OC, dyn/g, 3, 3], grouped
        OC, cin, g = self.out_dynamic, fc.in_channels, fc.groups
        cg, og = cin // g, (E * OC) // g
        dense = w.new_zeros((E * OC, cin, 3, 3))
        for grp in range(g):                                             # grouped filter bank -> dense rows (only routed rows run)
            dense[grp * og:(grp + 1) * og, grp * cg:(grp + 1) * cg] = w[grp * og:(grp + 1) * og]
        return {
            "st_dw": (ops.pack_dw_weight(dw_w, dtype), dw_b.contiguous()), "st_pw": (ops.pack_conv_weight(pw_w, dtype), pw_b.contiguous()),
            "cplx": _pack_conv(self.complexity_estimator[1], f32, device, pad_cout_to=4),
            "lin": _pack_conv(self.routing.router[0], f32, device, pad_cout_to=_ceil(E, 4)),
            "ew": ops.pack_conv_weight(dense, dtype).reshape(E, OC, -1).contiguous(),
            "en": (fe.expert_norm_weight.detach().float().to(device).contiguous(), fe.expert_norm_bias.detach().float().to(device).contiguous()),
            "proj": _pack_conv(self.proj, dtype, device), "bn": _pack_norm(self.bn, device), "consts": {},
        }

    def _run(self, x, out=None):
        B, H, W, C = x.shape
        pk = self._packed(x.device)
        st, dyn, E, k, ng, gs = self.static_channels, self.dynamic_channels, self.num_experts, self.top_k, self.num_groups, get_safe_groups
        xs, xd = x[..., :st], x[..., st:]
        s = ops.conv2d(ops.dwconv2d(xs, *pk["st_dw"], 3, True), *pk["st_pw"], 1, 1, True)
        cplx = ops.conv2d(ops.channel_stats(xd), *pk["cplx"], 1, 1, False)                                        # [B,1,1,4] fp32 logits
        lin = ops.conv2d(ops.channel_stats(xd, want_std=True), *pk["lin"], 1, 1, False)[..., :E]
        if B not in pk["consts"]:
            pk["consts"][B] = (torch.zeros((B, 1, 1, _ceil(E, 4)), dtype=torch.float32, device=x.device),
                               torch.full((B, 1, 1, 1), 100.0, dtype=torch.float32, device=x.device))
        g0, keep_all = pk["consts"][B]
        _, _, p0, _ = ops.gated_route_decide(g0[..., :E], lin, -100.0, 1.0, k, keep_all, clamp=0)                 # the router's own Softmax (gated.py:958): no clamp
        w, idx, probs, rows = ops.gated_route_decide(g0[..., :E], p0.view(B, 1, 1, E), -100.0, 1.0 / float(self.routing.temperature), k, keep_all,
                                                     clamp=2)                                                      # (probs / T).clamp(+-30) (gated.py:972)
        ops.batch_scale(w, cplx, 0.3, 1.5)                                                                         # routing_weights * complexity_scale
        self.last_route = {"weights": w, "indices": idx, "probs": probs}
        f = ops.expert_conv(xd, pk["ew"], 3, idx)
        f = ops.group_norm(f, gs(self.out_dynamic, ng), *pk["en"], 1e-5, act="silu", affine_rows=rows)
        d = ops.weighted_sum(w, [f[j * B:(j + 1) * B] for j in range(k)])
        cat = ops.channel_shuffle_cat([s, d], 1)
        return ops.group_norm(ops.conv2d(cat, *pk["proj"], 1, 1, False), gs(self.out_channels, ng), *pk["bn"], 1e-5, residual=x, out=out)


MIXTURE_BOUNDARY_MODULES = {**{c.__name__: c for c in GATED_CHAIN}, "SharedExpertMoE": SharedExpertMoE, "UltraOptimizedMoE": UltraOptimizedMoE, "ModularRouterExpertMoE": ModularRouterExpertMoE, "UltimateOptimizedMoE": UltimateOptimizedMoE, "OptimalHybridGateMoE": OptimalHybridGateMoE, "MultiHeadRouterMoE": MultiHeadRouterMoE, "DiversifiedExpertMoE": DiversifiedExpertMoE,
                            "GatedFusionMoE": GatedFusionMoE, "C2fMoA": C2fMoA, "C2fMoT": C2fMoT}
MIXTURE_BOUNDARY_REPEAT = {C2fMoA, C2fMoT}
