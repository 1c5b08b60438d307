"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the eval-mode forward of the Mixture-of-Transformer modules:
token-level top-k router, the three experts (LocalConv + GLU, Swin window + MLP, deformable 4-point + MLP), the
per-sample sparse dispatch of `_blend_experts`, output projection + GroupNorm + residual, and the `C2fMoT`
wrapper; spatial and image-level routers, optional scene-aware bias.  Every function cites the
reference lines it follows; parity is pinned by `tests/golden/make_golden_mot.py` (runs the REAL reference modules
on CPU) and checked without the reference by `tests/test_oracle_mot.py`.

State-dict layout: the reference's own parameter names under a module prefix `p`.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .moa_ref import _fp_floor, safe_groups
from .model_ref import _count, conv

NUM_EXPERTS = 3  # LocalConv / Window / Deformable (mot/block.py:53)


def _gn(sd, p, x, desired=8):
    w = sd[f"{p}.weight"]
    return F.group_norm(x, safe_groups(w.shape[0], desired), w, sd[f"{p}.bias"], 1e-5)


def _ln(sd, p, x):
    w = sd[f"{p}.weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[f"{p}.bias"], 1e-5)


def _lin(sd, p, x):
    return F.linear(x, sd[f"{p}.weight"], sd.get(f"{p}.bias"))


def sdpa(q, k, v, scale):
    """_sdpa (mot/experts.py:35-69) on torch >= 2.0."""
    return F.scaled_dot_product_attention(q, k, v, attn_mask=None, scale=scale)


# ---------------------------------------------------------------------------------- router
# Tie resolution for parity tests: {router prefix: bool [B, E, H, W]} — the selected experts per token are taken from the mask
# instead of this evaluation's own top-k (weights still come from this evaluation's probabilities).  Used when another fp32
# evaluation order resolved a near-tie between two experts' logits (gap < 1e-4) the other way: both resolutions are valid results
# of the reference algorithm, and everything downstream is then compared under the same resolution (tests/test_gpu_mixture.py).
FORCE_SELECT: dict = {}


def compute_scene_stats(x):
    """_MoTRouter.compute_scene_stats (mot/router.py:166-192): per image (high_frequency, heterogeneity, multi_scale) of the routed
    map in fp32 — mean |dx|, |dy| over the RMS; std / mean of the per-pixel channel energy; |var(pool 4x4) - var(pool 2x2)| over the
    variance (all biased)."""
    feature = x.float()
    eps = torch.finfo(feature.dtype).eps
    squared = feature.square()
    rms = squared.mean(dim=(1, 2, 3)).sqrt().clamp_min(eps)
    dx = (feature[..., 1:] - feature[..., :-1]).abs().mean(dim=(1, 2, 3)) if feature.shape[-1] > 1 else rms * 0
    dy = (feature[..., 1:, :] - feature[..., :-1, :]).abs().mean(dim=(1, 2, 3)) if feature.shape[-2] > 1 else rms * 0
    high_frequency = 0.5 * (dx + dy) / rms
    spatial_energy = squared.mean(dim=1)
    heterogeneity = spatial_energy.flatten(1).std(dim=1, unbiased=False) / spatial_energy.flatten(1).mean(dim=1).clamp_min(eps)
    pooled2 = F.adaptive_avg_pool2d(feature, (min(2, feature.shape[-2]), min(2, feature.shape[-1])))
    pooled4 = F.adaptive_avg_pool2d(feature, (min(4, feature.shape[-2]), min(4, feature.shape[-1])))
    scale2 = pooled2.var(dim=(1, 2, 3), unbiased=False)
    scale4 = pooled4.var(dim=(1, 2, 3), unbiased=False)
    multi_scale = (scale4 - scale2).abs() / feature.var(dim=(1, 2, 3), unbiased=False).clamp_min(eps)
    return torch.stack((high_frequency, heterogeneity, multi_scale), dim=1)


def mot_router(sd, p, x, top_k=2, use_spatial=True, scene_aware=False, scene_inference_mode="dynamic", info=None):
    """_MoTRouter.forward, eval (mot/router.py:118-136, 224-295): spatial = 1x1 -> GN(<=4) -> SiLU -> 1x1(+bias), image-level
    (use_spatial False) = GAP -> Linear(no bias) -> SiLU -> Linear, both in fp32; scene-aware routers in "dynamic" inference mode add
    scene_projector(compute_scene_stats(x)) = Linear(3, h) -> SiLU -> Linear(h, E) per image to the logits (:224-240);
    softmax(logits / T) with the persistent `temperature` buffer, hard top-k, stable_normalize over the selected
    set, scattered back to a sparse [B, E, H, W] (or [B, E, 1, 1]) weight map.  Returns (weights in x.dtype, indices, logits)."""
    xf = x.float()
    if use_spatial:
        h = F.conv2d(xf, sd[f"{p}.router.0.weight"].float())
        h = F.silu(_gn(sd, f"{p}.router.1", h, 4))
        logits = F.conv2d(h, sd[f"{p}.router.3.weight"].float(), sd[f"{p}.router.3.bias"].float()).float()
    else:
        h = F.silu(F.linear(F.adaptive_avg_pool2d(xf, 1).flatten(1), sd[f"{p}.router.2.weight"].float()))
        logits = F.linear(h, sd[f"{p}.router.4.weight"].float(), sd[f"{p}.router.4.bias"].float()).float().unsqueeze(-1).unsqueeze(-1)
    if scene_aware and scene_inference_mode == "dynamic":
        stats = compute_scene_stats(xf)
        hb = F.silu(F.linear(stats, sd[f"{p}.scene_projector.0.weight"].float(), sd[f"{p}.scene_projector.0.bias"].float()))
        bias = F.linear(hb, sd[f"{p}.scene_projector.2.weight"].float(), sd[f"{p}.scene_projector.2.bias"].float()).float()
        logits = logits + bias.unsqueeze(-1).unsqueeze(-1)
        if info is not None:
            info["scene_stats"], info["scene_bias"] = stats, bias
    w = F.softmax(logits / sd[f"{p}.temperature"].float(), dim=1)
    E = w.shape[1]
    if top_k < E:
        vals, idx = w.topk(top_k, dim=1)
        if p in FORCE_SELECT:
            idx = torch.where(FORCE_SELECT[p], w, torch.full_like(w, -1.0)).topk(top_k, dim=1).indices   # the forced set, ranked by this evaluation's probabilities
            vals = w.gather(1, idx)
        den = vals.sum(dim=1, keepdim=True).clamp_min(_fp_floor(1e-6, vals.dtype))   # stable_normalize (_numeric.py:85-90)
        w = torch.zeros_like(w).scatter_(1, idx, vals / den)
    else:
        idx = torch.arange(E).view(1, -1, 1, 1).expand(x.shape[0], -1, x.shape[2], x.shape[3])
    return w.to(x.dtype), idx, logits


# ---------------------------------------------------------------------------------- experts
def _pad_win(x, win):
    """[B, H, W, C] padded bottom/right to multiples of win (experts.py:237-245)."""
    B, H, W, C = x.shape
    ph, pw = (win - H % win) % win, (win - W % win) % win
    return F.pad(x, (0, 0, 0, pw, 0, ph)) if (ph or pw) else x


def _win_part(x, win):
    """[B, H, W, C] -> [B*nH*nW, win*win, C] (experts.py:247-252)."""
    B, H, W, C = x.shape
    return x.view(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, win * win, C)


def _win_rev(w, win, H, W):
    """[B*nH*nW, win*win, C] -> [B, H, W, C] (experts.py:254-261)."""
    B = w.shape[0] // ((H // win) * (W // win))
    return w.view(B, H // win, W // win, win, win, w.shape[2]).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def local_conv_expert(sd, p, x, nh, local_window_size=0):
    """_LocalConvTransformerExpert.forward (experts.py:123-171): GN -> DW3x3 -> 1x1 qkv, v += DW7x7(v), attention
    (global, or inside local windows when enabled and the map is larger than one window), 1x1 proj with layer
    scale; GN -> GLU FFN (sigmoid(Conv) * Conv -> Conv without act) with layer scale."""
    B, C, H, W = x.shape
    N, hd = H * W, C // nh
    xn = _gn(sd, f"{p}.norm1", x)
    qkv = F.conv2d(F.conv2d(xn, sd[f"{p}.dw_mix.weight"], None, 1, 1, 1, C), sd[f"{p}.qkv.weight"]).flatten(2)
    q, k, v = qkv.split(C, dim=1)
    v2 = v.reshape(B, C, H, W)
    v = (v2 + F.conv2d(v2, sd[f"{p}.pe.weight"], None, 1, 3, 1, C)).flatten(2)
    scale = hd ** -0.5
    if local_window_size > 0 and N > local_window_size ** 2:
        win = local_window_size
        qn, kn, vn = (_pad_win(t.reshape(B, C, H, W).permute(0, 2, 3, 1), win) for t in (q, k, v))
        Hp, Wp = qn.shape[1:3]
        qw, kw, vw = (_win_part(t, win).reshape(-1, win * win, nh, hd).permute(0, 2, 1, 3) for t in (qn, kn, vn))
        o = sdpa(qw, kw, vw, scale).transpose(1, 2).reshape(-1, win * win, C)
        o = _win_rev(o, win, Hp, Wp)[:, :H, :W, :].permute(0, 3, 1, 2).contiguous()
    else:
        qh, kh, vh = (t.view(B, nh, hd, N).transpose(2, 3) for t in (q, k, v))
        o = sdpa(qh, kh, vh, scale).transpose(2, 3).reshape(B, C, H, W)
    x = x + sd[f"{p}.ls1"] * F.conv2d(o, sd[f"{p}.proj.weight"])
    xn = _gn(sd, f"{p}.norm2", x)
    ffn = torch.sigmoid(conv(sd, f"{p}.ffn_gate.0", xn, fused=False)) * conv(sd, f"{p}.ffn_val", xn, fused=False)
    return x + sd[f"{p}.ls2"] * conv(sd, f"{p}.ffn_out", ffn, act=False, fused=False)


def window_expert(sd, p, x, nh, win=7, shift=0):
    """_WindowTransformerExpert.forward (experts.py:263-325): NHWC, pad to the window, optional cyclic shift by
    win//2 (odd blocks of the stack), LayerNorm -> per-window attention (Linear qkv / proj) -> un-shift, crop,
    layer-scale residual; LayerNorm -> Linear-GELU-Linear with layer scale."""
    B, C, H0, W0 = x.shape
    hd = C // nh
    x = _pad_win(x.permute(0, 2, 3, 1), win)
    H, W = x.shape[1], x.shape[2]
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    wins = _win_part(_ln(sd, f"{p}.norm1", x), win)
    Bw = wins.shape[0]
    qkv = _lin(sd, f"{p}.qkv", wins).reshape(Bw, win * win, 3, nh, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    a = sdpa(q, k, v, hd ** -0.5).transpose(1, 2).reshape(Bw, win * win, C)
    a = _win_rev(_lin(sd, f"{p}.proj", a), win, H, W)
    if shift > 0:
        a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    a = a[:, :H0, :W0, :]
    x = x[:, :H0, :W0, :]
    x = x + sd[f"{p}.ls1"] * a
    f = _lin(sd, f"{p}.ffn.3", F.gelu(_lin(sd, f"{p}.ffn.0", _ln(sd, f"{p}.norm2", x))))
    x = x + sd[f"{p}.ls2"] * f
    return x.permute(0, 3, 1, 2).contiguous()


def deformable_expert(sd, p, x, nh, n_points=4, align_corners=True):
    """_DeformableTransformerExpert.forward / _deform_attn (experts.py:381-493): per token and head, n_points
    sampling offsets (tanh, scaled by 0.25, clamped to the map) around the token's own normalised position and a
    softmax over the points; values bilinearly sampled (zeros padding) from the projected map; weighted sum,
    Linear out; layer-scale residual; LayerNorm -> Linear-GELU-Linear with layer scale."""
    B, C, H, W = x.shape
    N, hd = H * W, C // nh
    xf = x.flatten(2).transpose(1, 2)                     # [B, N, C]
    xn = _ln(sd, f"{p}.norm1", xf)
    q = _lin(sd, f"{p}.q_proj", xn)
    off = _lin(sd, f"{p}.offset_proj", q).reshape(B, N, nh, n_points, 2).tanh()
    aw = F.softmax(_lin(sd, f"{p}.attn_proj", q).reshape(B, N, nh, n_points), dim=-1)
    idx = torch.arange(N)
    row = (idx // W).float() / max(H - 1, 1) * 2 - 1
    col = (idx % W).float() / max(W - 1, 1) * 2 - 1
    ref = torch.stack([col, row], dim=-1)[None, :, None, None, :].expand(B, -1, nh, n_points, -1)
    locs = (ref + off * 0.25).clamp(-1.0, 1.0)
    v4 = _lin(sd, f"{p}.v_proj", xn).permute(0, 2, 1).reshape(B, C, H, W).reshape(B * nh, hd, H, W)
    locs = locs.permute(0, 2, 1, 3, 4).reshape(B * nh, N, n_points, 2)
    odt = v4.dtype
    if odt != torch.float32:
        v4, locs = v4.float(), locs.float()
    smp = F.grid_sample(v4, locs, mode="bilinear", padding_mode="zeros", align_corners=align_corners)
    if odt != torch.float32:
        smp = smp.to(odt)
    smp = smp.reshape(B, nh, hd, N, n_points).permute(0, 3, 1, 4, 2).contiguous()   # [B, N, nh, np, hd]
    a = _lin(sd, f"{p}.out_proj", (aw.unsqueeze(-1) * smp).sum(dim=3).reshape(B, N, C))
    xf = xf + sd[f"{p}.ls1"] * a
    f = _lin(sd, f"{p}.ffn.3", F.gelu(_lin(sd, f"{p}.ffn.0", _ln(sd, f"{p}.norm2", xf))))
    xf = xf + sd[f"{p}.ls2"] * f
    return xf.transpose(1, 2).reshape(B, C, H, W)


# ---------------------------------------------------------------------------------- block
def expert_heads(dim: int, num_heads: int) -> int:
    """MoTBlock.__init__ (mot/block.py:93-101): largest head count <= num_heads dividing dim."""
    h = num_heads
    while dim % h != 0 and h > 1:
        h -= 1
    return max(1, h)


def mot_block(sd, p, x, num_heads=8, top_k=2, window_size=7, n_points=4, window_shift=False, local_attn_window=0,
              grid_align_corners=True, info=None, use_spatial_router=True, scene_aware_router=False, scene_inference_mode="dynamic"):
    """MoTBlock.forward + _blend_experts, eval (mot/block.py:298-417): expert e runs on the images where at least
    one token selected it (experts in index order), its output is weighted per token, contributions accumulate in
    that order; then 1x1 out_proj -> GroupNorm(<=8) -> + x."""
    B = x.shape[0]
    nh = expert_heads(x.shape[1], num_heads)
    rinfo = {}
    w, idx, logits = mot_router(sd, f"{p}.router", x, top_k, use_spatial_router, scene_aware_router, scene_inference_mode, rinfo)
    if info is not None:
        info[p] = {"weights": w, "indices": idx, "logits": logits, **rinfo}
    fns = (
        lambda t: local_conv_expert(sd, f"{p}.experts.0", t, nh, local_attn_window),
        lambda t: window_expert(sd, f"{p}.experts.1", t, nh, window_size, window_size // 2 if window_shift else 0),
        lambda t: deformable_expert(sd, f"{p}.experts.2", t, nh, n_points, grid_align_corners),
    )
    out = x.new_zeros(x.shape)
    for e, fn in enumerate(fns):
        active = (idx == e).reshape(B, -1).any(dim=1)
        bi = torch.nonzero(active, as_tuple=True)[0]
        if bi.numel() == 0:
            continue
        out[bi] = out[bi] + (fn(x[bi]) * w[bi, e:e + 1]).to(out.dtype)
    out = _gn(sd, f"{p}.out_norm", F.conv2d(out, sd[f"{p}.out_proj.weight"]))
    return out + x


def c2f_heads(c: int, num_heads: int) -> int:
    """C2fMoT.__init__ (mot/wrappers.py:73-81): divides the width and keeps head_dim >= 8."""
    h = num_heads
    while h > 1 and (c % h != 0 or c // h < 8):
        h -= 1
    return max(1, h)


def c2f_mot(sd, p, x, num_heads=6, top_k=2, window_size=7, n_points=4, local_attn_window=0, info=None):
    """C2fMoT.forward (mot/wrappers.py:107-114): cv1 1x1 -> chunk(2) -> n MoTBlocks chained on the last chunk (odd
    blocks use shifted windows, :92) -> cat -> cv2 1x1."""
    y = list(conv(sd, f"{p}.cv1", x, fused=False).chunk(2, dim=1))
    heads = c2f_heads(y[0].shape[1], num_heads)
    for i in range(_count(sd, f"{p}.m")):
        y.append(mot_block(sd, f"{p}.m.{i}", y[-1], heads, top_k, window_size, n_points, bool(i % 2), local_attn_window,
                           True, info))
    return conv(sd, f"{p}.cv2", torch.cat(y, dim=1), fused=False)
