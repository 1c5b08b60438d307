"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the eval-mode forward of `UltimateOptimizedMoE` (moe/modules.py:1534-1700), the MoE block of the v0_3 master YAMLs:
channel split, static DW + PW path, `ZeroCostRouter` (moe/gated.py:938-992: a Linear over the [mean | std] channel statistics followed
by a Softmax INSIDE the router, then / temperature, clamp, a second softmax, top-k), batch-level complexity scale on the routing weights,
`FusedExpertGroup` (gated.py:1003-1098), 1x1 projection + GroupNorm + x.  Pinned bit for bit to the real reference by
tests/golden/make_golden_v03.py."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .gated_ref import _bn, _dw3, _gn, plain_fused_experts


def zero_cost_router(sd, p, x, top_k, temperature):
    """ZeroCostRouter.forward (gated.py:963-992), fp32."""
    B, C, H, W = x.shape
    xf = x.float()
    mean = xf.mean(dim=[2, 3])
    std = xf.std(dim=[2, 3], unbiased=False) if H * W > 1 else torch.zeros_like(mean)
    p0 = F.softmax(F.linear(torch.cat([mean, std], dim=1), sd[f"{p}.router.0.weight"]), dim=1)     # nn.Sequential(Linear, Softmax)
    logits = (p0 / temperature).clamp(-30.0, 30.0)
    probs = F.softmax(logits, dim=1)
    k = max(1, min(int(top_k), probs.shape[1]))
    tp, idx = torch.topk(probs, k, dim=1)
    tp = tp / (tp.sum(dim=1, keepdim=True) + 1e-6)
    return tp.to(x.dtype).view(B, k, 1, 1), idx.view(B, k, 1, 1), probs


def ultimate_optimized_moe(sd, p, x, num_experts=4, top_k=2, split_ratio=0.5, temperature=2.0, num_groups=8, info=None):
    """UltimateOptimizedMoE.forward, eval (modules.py:1650-1692).  `temperature`: the router's eval temperature = initial_temperature
    (2.0) for a model that has not been through the training schedule."""
    B, C, H, W = x.shape
    dyn = int(C * split_ratio)
    st = C - dyn
    xs, xd = x[:, :st], x[:, st:]
    c = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(xd, 1), sd[f"{p}.complexity_estimator.1.weight"], sd[f"{p}.complexity_estimator.1.bias"])).mean()
    c = torch.nan_to_num(c, nan=1.0, posinf=1.5, neginf=0.3).clamp(0.3, 1.5)
    s = F.silu(_bn(sd, f"{p}.static_net.1", _dw3(xs, sd[f"{p}.static_net.0.weight"])))
    s = F.silu(_bn(sd, f"{p}.static_net.4", F.conv2d(s, sd[f"{p}.static_net.3.weight"])))
    w, idx, probs = zero_cost_router(sd, f"{p}.routing", xd, top_k, temperature)
    w = w * c
    if info is not None:
        info[p] = {"weights": w, "indices": idx, "probs": probs, "complexity": c}
    d = plain_fused_experts(sd, f"{p}.fused_experts", xd, w, idx, num_experts, num_groups)
    out = F.conv2d(torch.cat([s, d], dim=1), sd[f"{p}.proj.weight"])
    return _gn(sd, f"{p}.bn", out, num_groups) + x
