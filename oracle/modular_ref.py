"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the eval-mode forward of `ModularRouterExpertMoE` (= `OptimizedMOEImproved`, moe/modules.py:957-1198, alias :1744),
the MoE block of the v0_1 master YAMLs (cfg/models/master/v0_1/det/*.yaml: the README's "YOLO-Master-v0.1" table), in its default
configuration — `EfficientSpatialRouter` (moe/routers.py:268-304 + BaseRouter._process_logits :185-255), `SimpleExpert`
(moe/experts.py:73-88) and the always-on shared expert.  Pinned bit for bit to the real reference by tests/golden/make_golden_v01.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .gated_ref import _bn, _gn


def efficient_spatial_router(sd, p, x, top_k, pool_scale=4):
    """EfficientSpatialRouter.forward, eval (routers.py:283-304): 4x4 average pool (when the map is larger than the pool) -> Conv3x3 ->
    BN -> SiLU -> Conv1x1 -> BN -> spatial mean in fp32; softmax in fp32, top-k, weights / clamp_min(sum, 1e-6) (routers.py:207-237)."""
    B, C, H, W = x.shape
    xin = F.avg_pool2d(x, kernel_size=pool_scale, stride=pool_scale) if (H > pool_scale and W > pool_scale) else x
    h = F.silu(_bn(sd, f"{p}.router.1", F.conv2d(xin, sd[f"{p}.router.0.weight"], padding=1)))
    out = _bn(sd, f"{p}.router.4", F.conv2d(h, sd[f"{p}.router.3.weight"]))
    logits = out.float().mean(dim=[2, 3])
    probs = F.softmax(logits.float(), dim=1)
    E = probs.shape[1]
    k = max(1, min(int(top_k), E))
    vals, idx = torch.topk(probs, k, dim=1)
    w = vals / vals.sum(dim=1, keepdim=True).clamp_min(1e-6)
    return w, idx, probs, logits


def simple_expert(sd, p, x, num_groups=8):
    """SimpleExpert.forward (experts.py:79-88): 1x1 -> GN -> SiLU -> 1x1 -> GN."""
    h = F.silu(_gn(sd, f"{p}.conv.1", F.conv2d(x, sd[f"{p}.conv.0.weight"]), num_groups))
    return _gn(sd, f"{p}.conv.4", F.conv2d(h, sd[f"{p}.conv.3.weight"]), num_groups)


def modular_router_expert_moe(sd, p, x, top_k=2, add_residual=True, info=None):
    """OptimizedMOEImproved.forward, eval (modules.py:1082-1168): route; shared expert (1x1 -> BN -> SiLU); for each expert in index
    order, the images that selected it run through it and are accumulated with their routing weight (fp32 accumulator for 16-bit
    inputs); shared + experts (+ x when the channel counts match)."""
    B = x.shape[0]
    E = 0
    while f"{p}.experts.{E}.conv.0.weight" in sd:
        E += 1
    w, idx, probs, logits = efficient_spatial_router(sd, f"{p}.routing", x, top_k)
    if info is not None:
        info[p] = {"weights": w, "indices": idx, "probs": probs, "logits": logits}
    shared = F.silu(_bn(sd, f"{p}.shared_expert.1", F.conv2d(x, sd[f"{p}.shared_expert.0.weight"])))
    cout = shared.shape[1]
    acc_dtype = torch.float32 if x.dtype in (torch.float16, torch.bfloat16) else x.dtype
    acc = torch.zeros(B, cout, x.shape[2], x.shape[3], dtype=acc_dtype)
    for i in range(E):
        mask = idx == i
        if mask.any():
            bi, ki = torch.where(mask)
            out = simple_expert(sd, f"{p}.experts.{i}", x[bi])
            acc.index_add_(0, bi, out.to(acc_dtype) * w[bi, ki].view(-1, 1, 1, 1).to(acc_dtype))
    y = (shared.to(acc_dtype) + acc).to(x.dtype)
    if add_residual and cout == x.shape[1]:
        y = y + x
    return y
