"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the eval-mode forward of `UltraOptimizedMoE` (moe/modules.py:121-232), the MoE block of
cfg/models/master/v0_1/det/yolo-master-n-uomoe*.yaml and exp/yolo-master-v0_2.yaml, in the configuration those YAMLs use
(`[c2, num_experts, top_k]`: `UltraEfficientRouter` moe/routers.py:58-147, `OptimizedSimpleExpert` moe/experts.py:13-29, the always-on
shared expert, `BatchedExpertComputation.compute_sparse_experts_batched` moe/utils.py:112-209).  Pinned bit for bit to the real reference
by tests/golden/make_golden_uomoe.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .gated_ref import _gn


def ultra_efficient_router(sd, p, x, top_k, pool_scale=8, temperature=1.0):
    """UltraEfficientRouter.forward, eval (routers.py:97-147): 8x8 average pool when the map is larger than the pool; DW3x3 -> GN(8) ->
    SiLU -> 1x1 -> GN(4) -> SiLU -> 1x1 (+bias); logits clamped to +-30, / temperature; softmax over the experts PER PIXEL in fp32, cast
    back to x's type; mean over the pixels; top-k of the pooled weights; values / clamp_min(sum, 1e-6)."""
    B, C, H, W = x.shape
    xd = F.avg_pool2d(x, kernel_size=pool_scale, stride=pool_scale) if (H > pool_scale and W > pool_scale) else x
    h = F.silu(_gn(sd, f"{p}.router.1", F.conv2d(xd, sd[f"{p}.router.0.weight"], padding=1, groups=C), 8))
    h = F.silu(_gn(sd, f"{p}.router.4", F.conv2d(h, sd[f"{p}.router.3.weight"]), 4))
    logits = F.conv2d(h, sd[f"{p}.router.6.weight"], sd[f"{p}.router.6.bias"])
    scaled = logits.clamp(-30.0, 30.0) / max(float(temperature), 1e-3)
    weights = F.softmax(scaled.float(), dim=1).type_as(x)
    pooled = weights.mean(dim=[2, 3], keepdim=True)
    E = pooled.shape[1]
    k = max(1, min(int(top_k), E))
    vals, idx = torch.topk(pooled, k, dim=1)
    vals = vals / vals.sum(dim=1, keepdim=True).clamp_min(1e-6)
    return vals.reshape(B, k), idx.reshape(B, k), pooled.reshape(B, E)


def optimized_simple_expert(sd, p, x, num_groups=8):
    """OptimizedSimpleExpert.forward (experts.py:28-29): 1x1 -> GN -> SiLU -> 1x1 -> GN."""
    h = F.silu(_gn(sd, f"{p}.conv.1", F.conv2d(x, sd[f"{p}.conv.0.weight"]), num_groups))
    return _gn(sd, f"{p}.conv.4", F.conv2d(h, sd[f"{p}.conv.3.weight"]), num_groups)


def ultra_optimized_moe(sd, p, x, top_k=2, num_groups=8, info=None):
    """UltraOptimizedMoE.forward, eval (modules.py:212-232): route; shared expert (1x1 -> GN -> SiLU); sparse experts — for every expert in
    index order the (image, slot) pairs that selected it with a weight ABOVE 0.01 (utils.py:166-169: the inference-only threshold) run
    through it, the fp32 product with the weight is accumulated into a zero tensor of x's type (index_add, utils.py:181-199), the sum
    clamped to +-1e4 (:203); shared + experts."""
    B = x.shape[0]
    E = 0
    while f"{p}.experts.{E}.conv.0.weight" in sd:
        E += 1
    w, idx, pooled = ultra_efficient_router(sd, f"{p}.routing", x, top_k)
    if info is not None:
        info[p] = {"weights": w, "indices": idx, "probs": pooled}
    shared = F.silu(_gn(sd, f"{p}.shared_expert.1", F.conv2d(x, sd[f"{p}.shared_expert.0.weight"]), num_groups))
    out = torch.zeros(B, shared.shape[1], x.shape[2], x.shape[3], dtype=x.dtype)
    valid = w > 0.01
    for e in range(E):
        mask = (idx == e) & valid
        if not mask.any():
            continue
        bi, ki = torch.where(mask)
        eo = optimized_simple_expert(sd, f"{p}.experts.{e}", x[bi], num_groups)
        out.index_add_(0, bi, (eo.float() * w[bi, ki].view(-1, 1, 1, 1).float()).to(out.dtype))
    out = out.clamp_(-1e4, 1e4)
    return shared + out
