"""Mixture-of-Transformer oracle against golden vectors produced by the REAL reference modules
(tests/golden/make_golden_mot.py).  CPU only; the reference is not needed at run time."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mot_ref

CASES = {
    "top2": dict(fn="block", kw={}),
    "shift": dict(fn="block", kw=dict(window_shift=True, local_attn_window=7)),
    "top1": dict(fn="block", kw=dict(top_k=1)),
    "dense": dict(fn="block", kw=dict(top_k=3)),
    "skip": dict(fn="block", kw={}),
    "c2f": dict(fn="c2f", kw={}),
    # round 4: scene-aware residual, image-level router, inference bypass (mot/router.py:118-136, 166-240)
    "scene": dict(fn="block", kw=dict(scene_aware_router=True)), "scene3": dict(fn="block", kw=dict(scene_aware_router=True, top_k=1)),
    "image": dict(fn="block", kw=dict(use_spatial_router=False)),
    "image_scene": dict(fn="block", kw=dict(use_spatial_router=False, scene_aware_router=True)),
    "scene_bypass": dict(fn="block", kw=dict(scene_aware_router=True, scene_inference_mode="bypass")),
}


def _load(golden_dir, name):
    z = np.load(golden_dir / f"mot_{name}.npz")
    sd = {f"m.{k}": torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}
    return sd, torch.from_numpy(z["x"]), torch.from_numpy(z["y"]), torch.from_numpy(z["router_w"]), torch.from_numpy(z["router_idx"])


@pytest.mark.parametrize("name", list(CASES))
def test_mot_oracle_reproduces_reference(name, golden_dir):
    sd, x, y, rw, ri = _load(golden_dir, name)
    info = {}
    with torch.inference_mode():
        if CASES[name]["fn"] == "block":
            out = mot_ref.mot_block(sd, "m", x, 6, info=info, **CASES[name]["kw"])
        else:
            out = mot_ref.c2f_mot(sd, "m", x, 6, info=info)
    # bit-identical at generation time (asserted by make_golden_mot.py); only the CPU thread count may differ here
    assert float((out - y).abs().max()) <= 1e-5 * float(y.abs().max()), f"max |dy| = {(out - y).abs().max().item():.3e}"
    w = torch.stack([v["weights"] for v in info.values()])
    idx = torch.stack([v["indices"] for v in info.values()])
    assert torch.equal(idx, ri), "top-k expert indices differ from the reference"
    assert float((w - rw).abs().max()) <= 1e-6
    k = CASES[name]["kw"].get("top_k", 2)
    z = np.load(golden_dir / f"mot_{name}.npz")
    if "scene_stats" in z.files:
        st = torch.stack([v["scene_stats"] for v in info.values()])
        assert float((st - torch.from_numpy(z["scene_stats"])).abs().max()) <= 1e-5 * float(st.abs().max())
        assert float((torch.stack([v["scene_bias"] for v in info.values()]) - torch.from_numpy(z["scene_bias"])).abs().max()) <= 1e-5
        assert float(torch.from_numpy(z["scene_bias"]).abs().max()) > 0.05, "the fixture's scene bias is too small to matter"
    else:
        assert all("scene_stats" not in v for v in info.values())
    assert torch.equal((w > 0).sum(2), torch.full_like((w > 0).sum(2), k))          # exactly top-k experts per token
    assert torch.allclose(w.sum(2), torch.ones_like(w.sum(2)), atol=1e-6)            # renormalised over the selected set
    if name == "skip":
        assert not bool((idx == 2).any()), "the deformable expert must never be selected in this fixture"


def test_mot_structural_properties(golden_dir):
    sd, x, _, _, _ = _load(golden_dir, "top2")
    nh = mot_ref.expert_heads(48, 6)
    with torch.inference_mode():
        # window expert: on a map that is a multiple of the window, shifting the INPUT cyclically by the shift and
        # running the unshifted expert equals the shifted expert up to the same cyclic shift of the output
        xs = x[:1, :, :14, :14].contiguous()
        a = mot_ref.window_expert(sd, "m.experts.1", xs, nh, 7, 3)
        b = torch.roll(mot_ref.window_expert(sd, "m.experts.1", torch.roll(xs, (-3, -3), (2, 3)), nh, 7, 0), (3, 3), (2, 3))
        assert torch.allclose(a, b, atol=1e-5)
        # deformable expert with zero offsets and uniform point weights samples each token's own value
        p = "m.experts.2"
        sd0 = dict(sd)
        for k in ("offset_proj.weight", "offset_proj.bias", "attn_proj.weight", "attn_proj.bias"):
            sd0[f"{p}.{k}"] = torch.zeros_like(sd[f"{p}.{k}"])
        got = mot_ref.deformable_expert(sd0, p, x, nh)
        xf = x.flatten(2).transpose(1, 2)
        xn = F.layer_norm(xf, (48,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"])
        att = F.linear(F.linear(xn, sd[f"{p}.v_proj.weight"]), sd[f"{p}.out_proj.weight"])
        xf = xf + sd[f"{p}.ls1"] * att
        f = F.linear(F.gelu(F.linear(F.layer_norm(xf, (48,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"]),
                                     sd[f"{p}.ffn.0.weight"], sd[f"{p}.ffn.0.bias"])), sd[f"{p}.ffn.3.weight"], sd[f"{p}.ffn.3.bias"])
        want = (xf + sd[f"{p}.ls2"] * f).transpose(1, 2).reshape(x.shape)
        assert torch.allclose(got, want, atol=2e-5)
    assert mot_ref.expert_heads(48, 6) == 6 and mot_ref.expert_heads(50, 8) == 5
    assert mot_ref.c2f_heads(48, 6) == 6 and mot_ref.c2f_heads(32, 6) == 4 and mot_ref.c2f_heads(20, 6) == 2
