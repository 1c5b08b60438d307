"""Mixture-of-Attention oracle (the next row of the hot path) against golden vectors produced by
the REAL reference modules (tests/golden/make_golden_moa.py).  CPU only; the reference is not needed at run time."""
import numpy as np
import pytest
import torch

from oracle import moa_ref

CASES = {
    "exact": dict(fn="block", kw={}),
    "blend": dict(fn="block", kw={}),
    "linear": dict(fn="block", kw={}),
    "kvcap": dict(fn="block", kw=dict(regional_max_kv_tokens=64, shortcut=False)),
    "c2f": dict(fn="c2f", kw={}),
    "sparse": dict(fn="block", kw=dict(sparse_inference=True, sparse_inference_threshold=0.2)),
    "sparse_one": dict(fn="block", kw=dict(sparse_inference=True, sparse_inference_threshold=0.99)),
}


def _load(golden_dir, name):
    z = np.load(golden_dir / f"moa_{name}.npz")
    sd = {f"m.{k}": torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()}
    return sd, torch.from_numpy(z["x"]), torch.from_numpy(z["y"]), torch.from_numpy(z["router_probs"])


@pytest.mark.parametrize("name", list(CASES))
def test_moa_oracle_reproduces_reference(name, golden_dir):
    sd, x, y, probs = _load(golden_dir, name)
    info = {}
    with torch.inference_mode():
        if CASES[name]["fn"] == "block":
            out = moa_ref.moa_block(sd, "m", x, 6, info=info, **CASES[name]["kw"])
        else:
            out = moa_ref.c2f_moa(sd, "m", x, 6, info=info)
    # bit-identical at generation time (asserted by make_golden_moa.py, same process as the reference); here only the
    # thread count of the CPU kernels may differ, which moves fp32 sums by an ulp or two
    assert float((out - y).abs().max()) <= 1e-5 * float(y.abs().max()), f"max |dy| = {(out - y).abs().max().item():.3e}"
    got = torch.stack([v["weights"] for v in info.values()])
    assert float((got - probs).abs().max()) <= 1e-6
    assert torch.allclose(got.sum(2), torch.ones_like(got.sum(2)), atol=1e-6)   # soft gate: sums to one per token


def test_moa_structural_properties(golden_dir):
    sd, x, _, _ = _load(golden_dir, "exact")
    B, C, H, W = x.shape
    nh, hd = 2, 16
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, nh, H * W, hd, generator=g) for _ in range(3))
    # a window that covers the whole (already aligned) map is plain attention
    full = moa_ref.sdpa(q[:, :, :14 * 14], k[:, :, :14 * 14], v[:, :, :14 * 14], hd ** -0.5)
    win = moa_ref.window_attn(q[:, :, :14 * 14], k[:, :, :14 * 14], v[:, :, :14 * 14], hd ** -0.5, 14, 14, 14)
    assert torch.allclose(full, win, atol=1e-6)
    # windows do not see each other: changing one window's keys leaves the others' outputs untouched
    k2 = k.clone()
    k2.view(B, nh, H, W, hd)[:, :, :7, :7] += 1.0
    a = moa_ref.window_attn(q, k, v, hd ** -0.5, 7, H, W).view(B, nh, H, W, hd)
    b = moa_ref.window_attn(q, k2, v, hd ** -0.5, 7, H, W).view(B, nh, H, W, hd)
    assert torch.equal(a[:, :, 7:, :], b[:, :, 7:, :]) and not torch.equal(a[:, :, :7, :7], b[:, :, :7, :7])
    # linear attention: output rows are convex-like combinations of v (positive features): bounded by v's range
    rf = sd["m.global_head._rf_matrix"]
    lo = moa_ref.linear_attn(q, k, v, rf)
    assert float(lo.max()) <= float(v.max()) + 1e-5 and float(lo.min()) >= float(v.min()) - 1e-5
    # head-count adjustment of the wrapper (divisible by 3, head_dim >= 16)
    assert moa_ref.effective_heads(48, 6) == 3 and moa_ref.effective_heads(128, 8) == 6 and moa_ref.effective_heads(256, 6) == 6
