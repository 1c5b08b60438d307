"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the eval-mode forward of the Mixture-of-Attention modules, the
next row of the hot path after the ES-MoE detector.  Dense soft routing (`sparse_inference=False`, the default of
the master YAMLs).  Every function cites the reference lines it follows; parity is pinned by
`tests/golden/make_golden_moa.py` (runs the REAL reference modules on CPU and stores golden vectors) and checked
without the reference by `tests/test_oracle_moa.py`.

State-dict layout: the reference's own parameter names under a module prefix `p` (e.g. `p.router.router.0.weight`).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .model_ref import _count, conv  # Conv (+BN, eval) restatement shared with the detector oracle

# ultralytics/nn/modules/moa/_constants.py
LINEAR_ATTN_THRESHOLD = 512
LINEAR_ATTN_BLEND_WINDOW = 64
LINEAR_ATTN_ACTIVATION_LIMIT = 1e4
NUM_GROUPS = 3  # local / regional / global (moa/block.py:47)


def safe_groups(channels: int, desired: int = 8) -> int:
    """Largest group count <= desired dividing channels (nn/modules/utils.py:108-115)."""
    if channels <= 0:
        return 1
    g = min(desired, channels)
    while channels % g:
        g -= 1
    return max(1, g)


def _fp_floor(value: float, dtype: torch.dtype) -> float:
    """nn/modules/_numeric.py:69-75."""
    if dtype == torch.float16:
        return max(value, 1e-4)
    if dtype == torch.bfloat16:
        return max(value, 1e-3)
    return value


def _gn(sd, p, x, desired):
    w = sd[f"{p}.weight"]
    return F.group_norm(x, safe_groups(w.shape[0], desired), w, sd[f"{p}.bias"], 1e-5)


def moa_router(sd, p, x, temperature: float = 1.0):
    """_MoARouter.forward (moa/router.py:50-62): 1x1 -> GroupNorm(<=4 groups) -> SiLU -> 1x1(+bias), logits / T,
    softmax over the 3 groups, all in fp32.  Returns (probs [B,3,H,W] in x.dtype, logits fp32)."""
    t = max(temperature, 0.1)  # router.py:38
    h = F.conv2d(x.float(), sd[f"{p}.router.0.weight"].float())
    h = F.silu(_gn(sd, f"{p}.router.1", h, 4))
    logits = F.conv2d(h, sd[f"{p}.router.3.weight"].float(), sd[f"{p}.router.3.bias"].float()).float() / t
    return F.softmax(logits, dim=1).to(x.dtype), logits


def sdpa(q, k, v, scale):
    """_flash_attn (moa/heads.py:35-52): torch's scaled_dot_product_attention with an explicit scale."""
    return F.scaled_dot_product_attention(q, k, v, scale=scale)


def _win_part(t, win):
    """[B, nh, H, W, hd] -> [B*nh*nW, win*win, hd] (heads.py:55-62)."""
    B, nh, H, W, hd = t.shape
    t = t.reshape(B, nh, H // win, win, W // win, win, hd)
    return t.permute(0, 1, 2, 4, 3, 5, 6).reshape(-1, win * win, hd)


def window_attn(q, k, v, scale, window_size, H, W):
    """_window_flash_attn (heads.py:83-117): zero-pad to a multiple of the window (bottom/right), attention inside
    each win x win window (padded tokens take part as zero keys/values), crop."""
    B, nh, N, hd = q.shape
    win = max(1, min(int(window_size), H, W))
    qs, ks, vs = (t.reshape(B, nh, H, W, hd) for t in (q, k, v))
    ph, pw = (win - H % win) % win, (win - W % win) % win
    if ph or pw:
        pad = (0, 0, 0, pw, 0, ph)
        qs, ks, vs = F.pad(qs, pad), F.pad(ks, pad), F.pad(vs, pad)
    Hp, Wp = qs.shape[2], qs.shape[3]
    o = sdpa(_win_part(qs, win), _win_part(ks, win), _win_part(vs, win), scale)
    o = o.reshape(B, nh, Hp // win, Wp // win, win, win, hd).permute(0, 1, 2, 4, 3, 5, 6).reshape(B, nh, Hp, Wp, hd)
    return o[:, :, :H, :W, :].reshape(B, nh, H * W, hd)


def _to_heads(t, B, nh, hd, N):
    return t.flatten(2).view(B, nh, hd, N).transpose(2, 3)  # [B, nh, N, hd]


def local_head(sd, p, x, nh, hd, window_size=7):
    """_LocalAttnHead.forward (heads.py:143-163): DW3x3 -> 1x1 qkv, v += DW7x7(v), window attention, 1x1 proj, GN(<=8)."""
    B, C, H, W = x.shape
    inner = nh * hd
    qkv = F.conv2d(F.conv2d(x, sd[f"{p}.qkv_dw.weight"], None, 1, 1, 1, C), sd[f"{p}.qkv_pw.weight"])
    q, k, v = qkv.split(inner, dim=1)
    v = v + F.conv2d(v, sd[f"{p}.pe.weight"], None, 1, 3, 1, inner)
    N = H * W
    o = window_attn(_to_heads(q, B, nh, hd, N), _to_heads(k, B, nh, hd, N), _to_heads(v, B, nh, hd, N), hd ** -0.5,
                    window_size, H, W)
    o = o.transpose(2, 3).reshape(B, inner, H, W)
    return _gn(sd, f"{p}.norm", F.conv2d(o, sd[f"{p}.proj.weight"]), 8)


def regional_head(sd, p, x, nh, hd, pool_stride=2, max_kv_tokens=4096):
    """_RegionalAttnHead.forward (heads.py:208-253): queries at full resolution, keys/values from an adaptive
    average pool (stride doubled until <= max_kv_tokens), exact attention, 1x1 proj, GN(<=8)."""
    B, C, H, W = x.shape
    inner = nh * hd
    if min(H, W) <= 1:
        kv = F.conv2d(x, sd[f"{p}.kv_proj.weight"])
    else:
        stride = pool_stride
        if max_kv_tokens is not None:
            while max(1, H // stride) * max(1, W // stride) > max_kv_tokens:
                stride *= 2
        kv = F.conv2d(F.adaptive_avg_pool2d(x, (max(1, H // stride), max(1, W // stride))), sd[f"{p}.kv_proj.weight"])
    n2 = kv.shape[2] * kv.shape[3]
    k, v = kv.split(inner, dim=1)
    k = k.flatten(2).view(B, nh, hd, n2).transpose(2, 3).contiguous()
    v = v.flatten(2).view(B, nh, hd, n2).transpose(2, 3).contiguous()
    q = F.conv2d(x, sd[f"{p}.q_proj.weight"]).flatten(2).view(B, nh, hd, H * W).transpose(2, 3).contiguous()
    o = sdpa(q, k, v, hd ** -0.5).transpose(2, 3).contiguous().reshape(B, inner, H, W)
    return _gn(sd, f"{p}.norm", F.conv2d(o, sd[f"{p}.proj.weight"]), 8)


def linear_attn(q, k, v, rf):
    """_GlobalAttnHead._linear_attn (heads.py:318-352): ReLU random-feature kernel attention, fp32 for half inputs."""
    B, nh, N, hd = q.shape
    odt = q.dtype
    if odt in (torch.float16, torch.bfloat16):
        q, k, v = q.float(), k.float(), v.float()
    rf = rf.to(dtype=q.dtype)
    nb = rf.shape[0]
    sc = nb ** -0.5
    eps = _fp_floor(1e-6, q.dtype)
    qf = (F.relu(q @ rf.T * sc) + eps).clamp(max=LINEAR_ATTN_ACTIVATION_LIMIT).reshape(B * nh, N, nb)
    kf = (F.relu(k @ rf.T * sc) + eps).clamp(max=LINEAR_ATTN_ACTIVATION_LIMIT).reshape(B * nh, N, nb)
    vf = v.reshape(B * nh, N, hd)
    kv = kf.transpose(1, 2) @ vf
    ksum = kf.float().sum(dim=1)
    numer = (qf @ kv).clamp(min=-LINEAR_ATTN_ACTIVATION_LIMIT, max=LINEAR_ATTN_ACTIVATION_LIMIT)
    denom = (qf @ ksum.to(qf.dtype).unsqueeze(-1)).clamp(min=_fp_floor(1e-6, qf.dtype))
    return (numer / denom).reshape(B, nh, N, hd).to(odt)


def global_head(sd, p, x, nh, hd):
    """_GlobalAttnHead.forward (heads.py:354-380): exact attention up to 512 tokens, linear blend with the
    random-feature attention inside (448, 512], random-feature attention beyond; 1x1 proj, GN(<=8).
    The orthogonal random-feature matrix is the persistent buffer `_rf_matrix` of the checkpoint."""
    B, C, H, W = x.shape
    N = H * W
    inner = nh * hd
    q, k, v = F.conv2d(x, sd[f"{p}.qkv.weight"]).flatten(2).split(inner, dim=1)
    q, k, v = (t.view(B, nh, hd, N).transpose(2, 3) for t in (q, k, v))
    rf = sd[f"{p}._rf_matrix"]
    if N <= LINEAR_ATTN_THRESHOLD:
        o = sdpa(q, k, v, hd ** -0.5)
        start = LINEAR_ATTN_THRESHOLD - LINEAR_ATTN_BLEND_WINDOW
        if N > start:
            alpha = (N - start) / LINEAR_ATTN_BLEND_WINDOW
            o = (1 - alpha) * o + alpha * linear_attn(q, k, v, rf)
    else:
        o = linear_attn(q, k, v, rf)
    o = o.transpose(2, 3).reshape(B, inner, H, W)
    return _gn(sd, f"{p}.norm", F.conv2d(o, sd[f"{p}.proj.weight"]), 8)


def moa_block(sd, p, x, num_heads, temperature=1.0, shortcut=True, local_window_size=7, regional_max_kv_tokens=4096,
              info=None, sparse_inference=False, sparse_inference_threshold=0.02):
    """MoABlock.forward (moa/block.py:167-278), eval: per-token softmax gate over the three heads (accumulated local,
    regional, global in that order), fusion Conv (no act), layer-scale residuals, FFN.  sparse_inference (block.py:194-234):
    a head group whose gate stays at or below the threshold for EVERY token of the batch is skipped (none above it: the group
    with the largest mean gate runs alone) and the retained gates are renormalised per token."""
    dim = x.shape[1]
    hd = max(dim // num_heads, 16)      # block.py:90
    nh = num_heads // NUM_GROUPS        # block.py:91
    w, logits = moa_router(sd, f"{p}.router", x, temperature)
    active = None
    if sparse_inference:
        active = w.amax(dim=(0, 2, 3)) > sparse_inference_threshold
        if not bool(active.any()):
            active = torch.zeros_like(active)
            active[w.mean(dim=(0, 2, 3)).argmax()] = True
        if bool(active.all()):
            active = None
    if info is not None:
        info[p] = {"weights": w, "logits": logits, "active": None if active is None else active.clone()}
    heads = (lambda: local_head(sd, f"{p}.local_head", x, nh, hd, local_window_size),
             lambda: regional_head(sd, f"{p}.region_head", x, nh, hd, 2, regional_max_kv_tokens),
             lambda: global_head(sd, f"{p}.global_head", x, nh, hd))
    if active is not None:
        bw = w * active.view(1, -1, 1, 1)
        bw = bw / bw.sum(dim=1, keepdim=True).clamp_min(torch.finfo(w.dtype).eps)
        mixed = x.new_zeros(x.shape)
        for g in range(NUM_GROUPS):
            if bool(active[g]):
                mixed = mixed + bw[:, g:g + 1] * heads[g]()
    else:
        mixed = w[:, 0:1] * heads[0]()
        mixed = mixed + w[:, 1:2] * heads[1]()
        mixed = mixed + w[:, 2:3] * heads[2]()
    mixed = conv(sd, f"{p}.fusion", mixed, act=False, fused=False)

    def ffn(t):
        return conv(sd, f"{p}.ffn.1", conv(sd, f"{p}.ffn.0", t, fused=False), act=False, fused=False)

    if shortcut:
        x = x + sd[f"{p}.ls_attn"] * mixed
        return x + sd[f"{p}.ls_ffn"] * ffn(x)
    x = sd[f"{p}.ls_attn"] * mixed
    return sd[f"{p}.ls_ffn"] * ffn(x)


def effective_heads(c: int, num_heads: int) -> int:
    """Head-count adjustment of C2fMoA.__init__ (moa/wrappers.py:95-121): divisible by 3, head_dim >= 16."""
    h = num_heads
    it = 256
    while h % NUM_GROUPS and it > 0:
        h += 1
        it -= 1
    it = 256
    while c // h < 16 and h > NUM_GROUPS and it > 0:
        h -= NUM_GROUPS
        it -= 1
    return max(h, NUM_GROUPS)


def c2f_moa(sd, p, x, num_heads=6, temperature=1.0, shortcut=True, local_window_size=7, regional_max_kv_tokens=4096,
            info=None):
    """C2fMoA.forward (moa/wrappers.py:144-177): cv1 1x1 -> chunk(2) -> n MoABlocks chained on the last chunk ->
    cat -> cv2 1x1."""
    y = list(conv(sd, f"{p}.cv1", x, fused=False).chunk(2, dim=1))
    c = y[0].shape[1]
    heads = effective_heads(c, num_heads)
    for i in range(_count(sd, f"{p}.m")):
        y.append(moa_block(sd, f"{p}.m.{i}", y[-1], heads, temperature, shortcut, local_window_size,
                           regional_max_kv_tokens, info))
    return conv(sd, f"{p}.cv2", torch.cat(y, dim=1), fused=False)
