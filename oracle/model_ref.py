"""ORACLE (test infrastructure — never imported by the product path).

CPU fp32 restatement of the reference's detection forward pass, written against the reference
source (paths relative to the Tencent/YOLO-Master checkout) and driven only by a model-YAML dict
and a ``state_dict`` with the reference's key names.  It uses plain ``torch.nn.functional`` ops
on CPU — the same third-party arithmetic (PyTorch aten/oneDNN) the reference itself runs on — so
that, for identical weights and inputs, it reproduces the reference bit-for-bit; this is pinned by
tests/golden/* (generated from the real reference by tests/golden/make_golden.py) and checked in
tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # initialize_weights sets eps=1e-3 on every BatchNorm2d (utils/torch_utils.py:552-562)

# Calibration mode (used only by tests/golden/make_calibration.py): when set to True, every BatchNorm's
# running statistics in `sd` are overwritten with the statistics of its actual input before being applied,
# which turns a random-weight network into a normalised one (what training would have produced).
CALIBRATE = False


def _bn(sd, p, y):
    """Eval-mode BatchNorm2d with parameters `p.{weight,bias,running_mean,running_var}`."""
    if CALIBRATE:
        sd[f"{p}.running_mean"] = y.mean(dim=(0, 2, 3)).clone()
        sd[f"{p}.running_var"] = y.var(dim=(0, 2, 3), unbiased=False).clamp_min(1e-4).clone()
    return F.batch_norm(y, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"],
                        False, 0.0, BN_EPS)


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


# ---------------------------------------------------------------------------------- primitives
def _fold(sd, p):
    """fuse_conv_and_bn (utils/torch_utils.py:315-349) for `p.conv` + `p.bn`."""
    w = sd[f"{p}.conv.weight"]
    co = w.shape[0]
    g, b = sd[f"{p}.bn.weight"], sd[f"{p}.bn.bias"]
    mu, var = sd[f"{p}.bn.running_mean"], sd[f"{p}.bn.running_var"]
    w_bn = torch.diag(g.div(torch.sqrt(BN_EPS + var)))
    wf = torch.mm(w_bn, w.view(co, -1)).view(w.shape)
    b_conv = torch.zeros(co)
    b_bn = b - g.mul(mu).div(torch.sqrt(var + BN_EPS))
    bf = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn
    return wf, bf


def conv(sd, p, x, k=1, s=1, act=True, g=1, fused=True):
    """Conv.forward_fuse / Conv.forward (nn/modules/conv.py:69-89); pad = k//2 (:30-36)."""
    if fused and not CALIBRATE:
        w, b = _fold(sd, p)
        y = F.conv2d(x, w, b, s, k // 2, 1, g)
    else:
        y = _bn(sd, f"{p}.bn", F.conv2d(x, sd[f"{p}.conv.weight"], None, s, k // 2, 1, g))
    return F.silu(y) if act else y


def _ksize(sd, p):
    return sd[f"{p}.conv.weight"].shape[-1]


def bottleneck(sd, p, x, fused=True):
    """Bottleneck.forward (nn/modules/block.py:484-486); shortcut iff c1 == c2."""
    k1, k2 = _ksize(sd, f"{p}.cv1"), _ksize(sd, f"{p}.cv2")
    y = conv(sd, f"{p}.cv2", conv(sd, f"{p}.cv1", x, k1, fused=fused), k2, fused=fused)
    add = sd[f"{p}.cv1.conv.weight"].shape[1] == sd[f"{p}.cv2.conv.weight"].shape[0]
    return x + y if add else y


def _count(sd, prefix):
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


def c3k(sd, p, x, fused=True):
    """C3.forward (nn/modules/block.py:349-351) with C3k's k=3 bottlenecks (:1114-1132)."""
    a = conv(sd, f"{p}.cv1", x, fused=fused)
    for j in range(_count(sd, f"{p}.m")):
        a = bottleneck(sd, f"{p}.m.{j}", a, fused)
    return conv(sd, f"{p}.cv3", torch.cat((a, conv(sd, f"{p}.cv2", x, fused=fused)), 1), fused=fused)


def c3k2(sd, p, x, fused=True):
    """C2f.forward (nn/modules/block.py:318-322) with C3k2's block choice (:1100-1111)."""
    y = list(conv(sd, f"{p}.cv1", x, fused=fused).chunk(2, 1))
    for j in range(_count(sd, f"{p}.m")):
        q = f"{p}.m.{j}"
        y.append(c3k(sd, q, y[-1], fused) if f"{q}.cv3.conv.weight" in sd else bottleneck(sd, q, y[-1], fused))
    return conv(sd, f"{p}.cv2", torch.cat(y, 1), fused=fused)


def aattn(sd, p, x, area, fused=True):
    """AAttn.forward (nn/modules/block.py:1696-1732), op for op."""
    B, _, H, W = x.shape
    N = H * W
    all_head_dim = sd[f"{p}.proj.conv.weight"].shape[1]
    head_dim = 32
    num_heads = all_head_dim // head_dim
    qkv = conv(sd, f"{p}.qkv", x, act=False, fused=fused).flatten(2).transpose(1, 2)
    if area > 1:
        qkv = qkv.reshape(B * area, N // area, all_head_dim * 3)
        B, N, _ = qkv.shape
    q, k, v = (qkv.view(B, N, num_heads, head_dim * 3).permute(0, 2, 3, 1)
               .split([head_dim, head_dim, head_dim], dim=2))
    attn = (q * (head_dim ** -0.5)).transpose(-2, -1) @ k
    attn = attn.softmax(dim=-1)
    o = v @ attn.transpose(-2, -1)
    o = o.permute(0, 3, 1, 2)
    v = v.permute(0, 3, 1, 2)
    if area > 1:
        o = o.reshape(B // area, N * area, all_head_dim)
        v = v.reshape(B // area, N * area, all_head_dim)
        B, N, _ = o.shape
    o = o.reshape(B, H, W, all_head_dim).permute(0, 3, 1, 2).contiguous()
    v = v.reshape(B, H, W, all_head_dim).permute(0, 3, 1, 2).contiguous()
    o = o + conv(sd, f"{p}.pe", v, 7, act=False, g=all_head_dim, fused=fused)
    return conv(sd, f"{p}.proj", o, act=False, fused=fused)


def ablock(sd, p, x, area, fused=True):
    """ABlock.forward (nn/modules/block.py:1787-1797)."""
    x = x + aattn(sd, f"{p}.attn", x, area, fused)
    return x + conv(sd, f"{p}.mlp.1", conv(sd, f"{p}.mlp.0", x, fused=fused), act=False, fused=fused)


def a2c2f(sd, p, x, area, fused=True):
    """A2C2f.forward (nn/modules/block.py:1865-1879), a2=True branch."""
    y = [conv(sd, f"{p}.cv1", x, fused=fused)]
    for i in range(_count(sd, f"{p}.m")):
        h = y[-1]
        for j in range(_count(sd, f"{p}.m.{i}")):
            h = ablock(sd, f"{p}.m.{i}.{j}", h, area, fused)
        y.append(h)
    out = conv(sd, f"{p}.cv2", torch.cat(y, 1), fused=fused)
    if f"{p}.gamma" in sd:
        g = sd[f"{p}.gamma"]
        return x + g.view(-1, g.shape[0], 1, 1) * out
    return out


# ---------------------------------------------------------------------------------- ES-MoE
def stable_normalize(t, dim, eps=1e-6):
    """nn/modules/_numeric.py:85-90 (fp32)."""
    return t / t.sum(dim=dim, keepdim=True).clamp_min(eps)


def esmoe_route(sd, p, x, top_k=2, thr=0.4, hard_top_k=True, sparse=True):
    """DynamicRoutingLayer.forward + _hard_top_k (moe/routers.py:458-527) and the dispatch decision of
    ES_MOE._sparse_forward (moe/modules.py:665-684).  Returns (route_w [B,E], gate_w [B,E], retained [B,E], logits).
    hard_top_k=False: `top_k=None` modules, plain softmax (routers.py:477-479).  sparse=False (or top_k >= E): the dense
    forward (modules.py:648-656) — every expert is summed with its routing weight, nothing is pruned or renormalised;
    `retained` then marks the experts with a non-zero weight (the others contribute exactly 0)."""
    B = x.shape[0]
    pooled = F.adaptive_avg_pool2d(x, 1)
    h = F.silu(F.conv2d(pooled, sd[f"{p}.routing.routing_network.0.weight"], sd[f"{p}.routing.routing_network.0.bias"]))
    logits = F.conv2d(h, sd[f"{p}.routing.routing_network.2.weight"], sd[f"{p}.routing.routing_network.2.bias"])
    E = logits.shape[1]
    w = F.softmax(logits.reshape(B, E, -1).float().clamp(-30.0, 30.0), dim=1)
    if hard_top_k:
        values, indices = torch.topk(w, top_k, dim=1)
        values = stable_normalize(values, dim=1)
        sparse_w = torch.zeros_like(w)
        sparse_w.scatter_(1, indices, values)
        route_w = sparse_w.view(B, E)
    else:
        route_w, top_k = w.view(B, E), E
    if top_k >= E or not sparse:
        return route_w, route_w.clone(), route_w > 0, logits.view(B, E)
    importance = route_w  # mean over H*W of a spatially constant map (modules.py:668-669)
    topv, topi = torch.topk(importance, top_k, dim=1)
    keep = torch.ones_like(topi, dtype=torch.bool)
    if thr > 0:
        ranks = torch.arange(top_k).view(1, -1)
        keep = (ranks == 0) | (topv >= thr)
    retained = torch.zeros(B, E, dtype=torch.bool)
    retained.scatter_(1, topi, keep)
    rw = route_w * retained.to(route_w.dtype)
    gate_w = rw / rw.sum(dim=1, keepdim=True).clamp_min(torch.finfo(torch.float32).eps)
    return route_w, gate_w, retained, logits.view(B, E)


def esmoe_state(route_w, H=1, W=1):
    """Eval-time buffers ES_MOE keeps (moe/modules.py:706-741, moe/loss.py:16-26): `expert_usage_counts` = mean routing
    weight per expert over the batch (the weight map is spatially constant), `load_balancing_loss` = E * sum(u_n^2) with
    u_n = usage / clamp_min(sum usage, 1e-6)."""
    B, E = route_w.shape
    usage = route_w.view(B, E, 1, 1).repeat(1, 1, H, W).mean(dim=(0, 2, 3))   # the reference averages the repeated map
    un = usage.reshape(-1).float()
    un = un / un.sum().clamp_min(1e-6)
    return usage, E * torch.sum(un * un)


def expert(sd, p, x):
    """DepthwiseSeparableConv.forward (moe/experts.py:291-296): DW -> PW -> BN -> SiLU (never BN-folded)."""
    k = sd[f"{p}.conv.depthwise.weight"].shape[-1]
    y = F.conv2d(x, sd[f"{p}.conv.depthwise.weight"], None, 1, (k - 1) // 2, 1, x.shape[1])
    y = F.conv2d(y, sd[f"{p}.conv.pointwise.weight"])
    return F.silu(_bn(sd, f"{p}.conv.bn", y))


def es_moe(sd, p, x, top_k=2, thr=0.4, info=None, sparse=True, hard_top_k=True):
    """ES_MOE.forward, eval (moe/modules.py:535-583): sparse dispatch (:659-704) or the dense sum (:648-656)."""
    B, C, H, W = x.shape
    route_w, gate_w, retained, logits = esmoe_route(sd, p, x, top_k, thr, hard_top_k, sparse)
    E = route_w.shape[1]
    co = sd[f"{p}.norm.0.weight"].shape[0]
    if sparse and hard_top_k and top_k < E:
        out = x.new_zeros(B, co, H, W)
        for e in range(E):
            idx = torch.where(retained[:, e])[0]
            if idx.numel() == 0:
                if CALIBRATE:
                    expert(sd, f"{p}.experts.{e}", x)  # give unselected experts sane statistics too
                continue
            eo = expert(sd, f"{p}.experts.{e}", x[idx])
            out.index_add_(0, idx, eo * gate_w[idx, e].view(-1, 1, 1, 1))
    else:
        out = 0
        for e in range(E):
            out = out + expert(sd, f"{p}.experts.{e}", x) * route_w[:, e].view(B, 1, 1, 1).repeat(1, 1, H, W)
    out = _bn(sd, f"{p}.norm.0", out)
    if info is not None:
        usage, lb = esmoe_state(route_w, H, W)
        info[p] = {"route_w": route_w, "gate_w": gate_w, "retained": retained, "logits": logits, "usage": usage, "lb_loss": lb}
    return F.silu(out)


# ---------------------------------------------------------------------------------- Detect
def detect(sd, p, feats, strides, nc=80, reg_max=16, fused=True):
    """Detect.forward eval (nn/modules/head.py:146-194), legacy=False head; returns (y, boxes, scores)."""
    bs = feats[0].shape[0]
    boxes, scores = [], []
    for i, x in enumerate(feats):
        b = conv(sd, f"{p}.cv2.{i}.1", conv(sd, f"{p}.cv2.{i}.0", x, 3, fused=fused), 3, fused=fused)
        b = F.conv2d(b, sd[f"{p}.cv2.{i}.2.weight"], sd[f"{p}.cv2.{i}.2.bias"])
        if f"{p}.cv3.{i}.0.0.conv.weight" in sd:  # DW3x3 + 1x1, twice (head.py:111-118)
            c = conv(sd, f"{p}.cv3.{i}.0.0", x, 3, g=x.shape[1], fused=fused)
            c = conv(sd, f"{p}.cv3.{i}.0.1", c, 1, fused=fused)
            c = conv(sd, f"{p}.cv3.{i}.1.0", c, 3, g=c.shape[1], fused=fused)
            c = conv(sd, f"{p}.cv3.{i}.1.1", c, 1, fused=fused)
        else:  # legacy: Conv3x3, Conv3x3
            c = conv(sd, f"{p}.cv3.{i}.1", conv(sd, f"{p}.cv3.{i}.0", x, 3, fused=fused), 3, fused=fused)
        c = F.conv2d(c, sd[f"{p}.cv3.{i}.2.weight"], sd[f"{p}.cv3.{i}.2.bias"])
        boxes.append(b.view(bs, 4 * reg_max, -1))
        scores.append(c.view(bs, nc, -1))
    boxes, scores = torch.cat(boxes, -1), torch.cat(scores, -1)
    # make_anchors (utils/tal.py:398-411)
    ap, st = [], []
    for x, s in zip(feats, strides):
        h, w = x.shape[2:]
        sx = torch.arange(end=w, dtype=torch.float32) + 0.5
        sy = torch.arange(end=h, dtype=torch.float32) + 0.5
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        ap.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=torch.float32))
    anchors, stride_t = torch.cat(ap).transpose(0, 1), torch.cat(st).transpose(0, 1)
    # DFL (nn/modules/block.py:81-84) + dist2bbox xywh (utils/tal.py:414-423)
    b, _, a = boxes.shape
    dflw = sd.get(f"{p}.dfl.conv.weight", torch.arange(reg_max, dtype=torch.float32).view(1, reg_max, 1, 1))
    dist = F.conv2d(boxes.view(b, 4, reg_max, a).transpose(2, 1).softmax(1), dflw).view(b, 4, a)
    lt, rb = dist.chunk(2, 1)
    x1y1 = anchors.unsqueeze(0) - lt
    x2y2 = anchors.unsqueeze(0) + rb
    dbox = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], 1) * stride_t
    return torch.cat((dbox, scores.sigmoid()), 1), boxes, scores


# ---------------------------------------------------------------------------------- Segment
def proto(sd, p, x, fused=True):
    """Proto.forward (nn/modules/block.py:88-107): Conv3x3 -> ConvTranspose2d(2, 2, bias) -> Conv3x3 -> Conv1x1."""
    h = conv(sd, f"{p}.cv1", x, 3, fused=fused)
    h = F.conv_transpose2d(h, sd[f"{p}.upsample.weight"], sd[f"{p}.upsample.bias"], stride=2)
    return conv(sd, f"{p}.cv3", conv(sd, f"{p}.cv2", h, 3, fused=fused), 1, fused=fused)


def segment(sd, p, feats, strides, nc=80, nm=32, reg_max=16, fused=True):
    """Segment.forward eval (nn/modules/head.py:317-349): Detect's output with the mask coefficients of the three cv4
    branches (Conv3x3, Conv3x3, 1x1 + bias) appended as rows [4+nc, 4+nc+nm), and the prototype maps of level 0.
    Returns (y [B, 4+nc+nm, A], boxes, scores, mc [B, nm, A], proto [B, nm, 2H0, 2W0])."""
    y, boxes, scores = detect(sd, p, feats, strides, nc, reg_max, fused)
    bs = feats[0].shape[0]
    mc = []
    for i, x in enumerate(feats):
        c = conv(sd, f"{p}.cv4.{i}.1", conv(sd, f"{p}.cv4.{i}.0", x, 3, fused=fused), 3, fused=fused)
        mc.append(F.conv2d(c, sd[f"{p}.cv4.{i}.2.weight"], sd[f"{p}.cv4.{i}.2.bias"]).view(bs, nm, -1))
    mc = torch.cat(mc, 2)
    return torch.cat([y, mc], 1), boxes, scores, mc, proto(sd, f"{p}.proto", feats[0], fused)


# ---------------------------------------------------------------------------------- graph
def forward(cfg: dict, sd: dict, x: torch.Tensor, fused: bool = True, taps: dict | None = None,
            moe_info: dict | None = None):
    """Run the YAML graph (BaseModel._predict_once, nn/tasks.py:182-218).  cfg: model-YAML dict with
    ``scale``; sd: reference-keyed state_dict (fp32, CPU).  Returns (y [B,4+nc,A], boxes, scores).
    ``taps[i]`` receives every layer's NCHW output."""
    rows = cfg["backbone"] + cfg["head"]
    nc = cfg["nc"]
    ys, cur = [], x
    save = set()
    for i, (f, n, m, args) in enumerate(rows):
        for j in ([f] if isinstance(f, int) else f):
            if j != -1:
                save.add(j % i)
    # SharedExpertMoE (moe/shared_expert_moe.py:85-115): every block of a pool aliases ONE expert-group module, listed in the state_dict
    # under each member's prefix; Module.load_state_dict walks the children in order and copies into the same tensors every time, so the
    # entries of the pool's LAST member are the weights every member computes with.  Restated on the dict: members read the last member's keys.
    pools = {}
    for i, (f, n, m, args) in enumerate(rows):
        if m == "SharedExpertMoE":
            pools.setdefault(str(args[13]) if len(args) > 13 else "shared", []).append(i)
    if any(len(v) > 1 for v in pools.values()):
        sd = dict(sd)
        for members in pools.values():
            last = f"model.{members[-1]}.fused_experts."
            for i in members[:-1]:
                for k in [k for k in sd if k.startswith(last)]:
                    sd[f"model.{i}.fused_experts." + k[len(last):]] = sd[k]
    scale_of = []
    out = None
    for i, (f, n, m, args) in enumerate(rows):
        p = f"model.{i}"
        if f != -1:
            cur = ys[f] if isinstance(f, int) else [cur if j == -1 else ys[j] for j in f]
        if isinstance(f, int):
            sc = (scale_of[-1] if scale_of else 1.0) if f == -1 else scale_of[f]
        else:
            sc = scale_of[-1] if f[0] == -1 else scale_of[f[0]]
        if m == "Conv":
            k = args[1] if len(args) > 1 else 1
            s = args[2] if len(args) > 2 else 1
            cur = conv(sd, p, cur, k, s, fused=fused)
            sc *= s
        elif m == "C3k2":
            cur = c3k2(sd, p, cur, fused)
        elif m == "A2C2f":
            cur = a2c2f(sd, p, cur, args[2] if len(args) > 2 else 1, fused)
        elif m == "ES_MOE":
            cur = es_moe(sd, p, cur, info=moe_info)
        elif m == "VisualEnhancedAdaptiveGateMoE":   # args after width scaling: [c2, num_experts, top_k, split_ratio]
            from . import gated_ref
            cur = gated_ref.visual_enhanced_moe(sd, p, cur, num_experts=args[1], top_k=args[2],
                                                split_ratio=args[3] if len(args) > 3 else 0.5, info=moe_info)
        elif m == "SharedExpertMoE":   # v0_8 shared-pool rows: [c2, E, k, split, groups, T0, T1, 3 loss coefficients, fused threshold, shuffle groups, bottleneck, pool_id]
            from . import gated_ref
            cur = gated_ref.adaptive_gate_chain(sd, p, cur, num_experts=args[1], top_k=args[2], split_ratio=args[3] if len(args) > 3 else 0.5,
                                                num_groups=args[4] if len(args) > 4 else 8, temperature=args[5] if len(args) > 5 else 1.2,
                                                shuffle_groups=args[11] if len(args) > 11 else 2, complexity_after_hooks=False, info=moe_info)
        elif m in ("AdaptiveGateMoE", "FusedAdaptiveGateMoE", "HybridAdaptiveGateMoE", "HybridAdaptiveGateMoEv2", "LowRankHybridAdaptiveGateMoE",
                   "RefinedLowRankHybridAdaptiveGateMoE", "DetailAwareLowRankHybridAdaptiveGateMoE",
                   "ContextRefinedLowRankHybridAdaptiveGateMoE"):      # v0_4 ... v0_9 / v0_11 rows: [c2, num_experts, top_k, split_ratio]
            from . import gated_ref
            plain = m in ("AdaptiveGateMoE", "FusedAdaptiveGateMoE")    # constructor default temperature 1.0, no channel shuffle
            cur = gated_ref.adaptive_gate_chain(sd, p, cur, num_experts=args[1], top_k=args[2], split_ratio=args[3] if len(args) > 3 else 0.5,
                                                temperature=1.0 if plain else 1.2, shuffle_groups=1 if plain else 2,
                                                complexity_after_hooks=not plain, info=moe_info)
        elif m in ("OptimalHybridGateMoE", "GatedFusionMoE", "MultiHeadRouterMoE", "DiversifiedExpertMoE"):   # v0_12 / v0_15 rows: [c2, num_experts, top_k, split_ratio]
            from . import gated_ref
            cur = gated_ref.optimal_hybrid_moe(sd, p, cur, num_experts=args[1], top_k=args[2],
                                               split_ratio=args[3] if len(args) > 3 else 0.5, cross_gate=m == "GatedFusionMoE",
                                               info=moe_info)
        elif m == "UltimateOptimizedMoE":            # v0_3 rows: [c2, num_experts, top_k, split_ratio]
            from . import ultimate_ref
            cur = ultimate_ref.ultimate_optimized_moe(sd, p, cur, num_experts=args[1], top_k=args[2], split_ratio=args[3] if len(args) > 3 else 0.5,
                                                      info=moe_info)
        elif m == "UltraOptimizedMoE":               # v0_1 uomoe / exp v0_2 rows: [c2, num_experts, top_k]
            from . import uomoe_ref
            cur = uomoe_ref.ultra_optimized_moe(sd, p, cur, top_k=args[2] if len(args) > 2 else 2, info=moe_info)
        elif m == "ModularRouterExpertMoE":          # v0_1 rows: [c2, num_experts, top_k]
            from . import modular_ref
            cur = modular_ref.modular_router_expert_moe(sd, p, cur, top_k=args[2] if len(args) > 2 else 2, info=moe_info)
        elif m == "C2fMoA":                          # [c2, num_heads, mlp_ratio, temperature, shortcut]
            from . import moa_ref
            cur = moa_ref.c2f_moa(sd, p, cur, num_heads=args[1] if len(args) > 1 else 6,
                                  temperature=args[3] if len(args) > 3 else 1.0,
                                  shortcut=args[4] if len(args) > 4 else True, info=moe_info)
        elif m == "C2fMoT":                          # [c2, num_heads, top_k, window_size, n_points, mlp_ratio, T, coeff, e]
            from . import mot_ref
            cur = mot_ref.c2f_mot(sd, p, cur, num_heads=args[1] if len(args) > 1 else 6, top_k=args[2] if len(args) > 2 else 2,
                                  window_size=args[3] if len(args) > 3 else 7, n_points=args[4] if len(args) > 4 else 4,
                                  info=moe_info)
        elif m == "nn.Upsample":
            cur = F.interpolate(cur, scale_factor=2.0, mode="nearest")
            sc /= 2
        elif m == "Concat":
            cur = torch.cat(cur, 1)
        elif m == "Detect":
            out = detect(sd, p, cur, [scale_of[j] for j in f], nc, cfg.get("reg_max", 16), fused)
            cur = out[0]
        elif m == "Segment":                         # [nc, nm, npr]; out = (y with mask rows, boxes, scores, mc, proto)
            out = segment(sd, p, cur, [scale_of[j] for j in f], nc, args[1] if len(args) > 1 else 32, cfg.get("reg_max", 16), fused)
            cur = out[0]
        else:
            raise KeyError(m)
        scale_of.append(sc)
        ys.append(cur if i in save else None)
        if taps is not None:
            taps[i] = cur
    return out
