"""Golden vectors for the Mixture-of-Attention oracle (oracle/moa_ref.py), produced by the REAL reference modules
(`ultralytics.nn.modules.moa`) on CPU.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_moa.py

Writes tests/golden/moa_<case>.npz: the seeded state_dict, the input, the reference output and router
probabilities.  The script also asserts that the oracle reproduces the reference bit for bit on every case.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import moa_ref, refboot  # noqa: E402

refboot.boot()  # stubs cv2 and puts /root/reference on sys.path

from ultralytics.nn.modules.moa.block import MoABlock  # noqa: E402
from ultralytics.nn.modules.moa.wrappers import C2fMoA  # noqa: E402


def seeded_fill(module, seed):
    """Deterministic non-degenerate parameters (the reference init zeroes the router head and uses std 0.02)."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    out = {}
    for k, v in sd.items():
        if not v.is_floating_point() or k.endswith("_rf_matrix") or v.dim() == 0:   # counters, fixed bases, temperature
            out[k] = v.clone()
        elif k.endswith("running_var"):
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith(".bias"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif "ls_attn" in k or "ls_ffn" in k:
            out[k] = 0.5 + 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1:  # norm scales
            out[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:               # conv weights: unit-gain fan-in scaling
            fan_in = v[0].numel()
            out[k] = torch.randn(v.shape, generator=g) * (1.5 / fan_in ** 0.5)
    module.load_state_dict(out)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3  # what initialize_weights sets inside a DetectionModel (utils/torch_utils.py:552-562)
    return out


def case(name, module, oracle_fn, x, seed):
    sd = seeded_fill(module, seed)
    module.eval()
    with torch.inference_mode():
        y = module(x)
        info = {}
        oy = oracle_fn({f"m.{k}": v for k, v in sd.items()}, x, info)
    exact = torch.equal(y, oy)
    print(f"[moa_{name}] x {tuple(x.shape)} -> y {tuple(y.shape)}; oracle bit-exact vs reference: {exact}; "
          f"max|dy| {(y - oy).abs().max().item():.3e}; |y| max {y.abs().max().item():.3f}")
    assert exact
    probs = np.stack([v["weights"].numpy() for v in info.values()])
    rec = {"x": x.numpy(), "y": y.numpy(), "router_probs": probs, "keys": np.array(list(sd.keys()))}
    rec.update({f"sd::{k}": v.numpy() for k, v in sd.items()})
    np.savez_compressed(HERE / f"moa_{name}.npz", **rec)


def sparse_cases():
    """`sparse_inference=True` (moa/block.py:194-234; `python tests/golden/make_golden_moa.py sparse`): the router's last bias pushes one
    group below the threshold for every token (`sparse`: the global head is skipped, the other two renormalised) / every group below it
    (`sparse_one`: threshold 0.99, the group with the largest mean gate runs alone)."""
    g3 = torch.Generator().manual_seed(77)
    for name, thr, bias in (("sparse", 0.2, (0.0, 0.0, -6.0)), ("sparse_one", 0.99, (0.5, 0.0, -0.5))):
        m = MoABlock(48, num_heads=6, sparse_inference=True, sparse_inference_threshold=thr)
        x = torch.randn(2, 48, 14, 18, generator=g3)

        def fill(module, seed, bias=bias):
            sd = seeded_fill(module, seed)
            sd["router.router.3.bias"] = sd["router.router.3.bias"] + torch.tensor(bias)
            sd["router.router.3.weight"] = sd["router.router.3.weight"] * 0.3      # gates near the biases: the decision is not a near call
            module.load_state_dict(sd)
            return sd
        sd = fill(m, 8)
        m.eval()
        with torch.inference_mode():
            y = m(x)
            info = {}
            oy = moa_ref.moa_block({f"m.{k}": v for k, v in sd.items()}, "m", x, 6, info=info, sparse_inference=True, sparse_inference_threshold=thr)
        act = info["m"]["active"]
        snap = m.last_routing_snapshot
        print(f"[moa_{name}] active {None if act is None else act.tolist()} executed_groups {snap['executed_groups']} dropped mass {snap['dropped_routing_mass']:.4f}; "
              f"oracle bit-exact vs reference: {torch.equal(y, oy)}")
        assert torch.equal(y, oy) and act is not None and int(act.sum()) == snap["executed_groups"]
        rec = {"x": x.numpy(), "y": y.numpy(), "router_probs": info["m"]["weights"].numpy()[None], "keys": np.array(list(sd.keys())),
               "active": act.numpy(), "threshold": np.float32(thr), "dropped_routing_mass": np.float32(snap["dropped_routing_mass"])}
        rec.update({f"sd::{k}": v.numpy() for k, v in sd.items()})
        np.savez_compressed(HERE / f"moa_{name}.npz", **rec)


def extra_cases():
    """Cases added after the first fixture set; each has its own generator so that re-running never touches the others.
    `python tests/golden/make_golden_moa.py extra`."""
    g2 = torch.Generator().manual_seed(2024)
    # the block of BASELINE config 5 (L scale): c = 128, 6 heads -> head_dim = 21 (not a multiple of 8), 2 heads per group
    m = MoABlock(128, num_heads=6)
    case("hd21", m, lambda sd, xx, info: moa_ref.moa_block(sd, "m", xx, 6, info=info), torch.randn(2, 128, 9, 13, generator=g2), 7)


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        extra_cases()
        raise SystemExit
    if len(sys.argv) > 1 and sys.argv[1] == "sparse":
        sparse_cases()
        raise SystemExit
    g = torch.Generator().manual_seed(123)

    def blk(x, seed, name, **kw):
        m = MoABlock(48, num_heads=6, **kw)
        case(name, m, lambda sd, xx, info: moa_ref.moa_block(sd, "m", xx, 6, info=info, **{k: v for k, v in kw.items()
                                                                                            if k in ("shortcut", "local_window_size", "regional_max_kv_tokens")}), x, seed)

    blk(torch.randn(2, 48, 14, 18, generator=g), 1, "exact")            # N = 252: exact global attention, padded windows
    blk(torch.randn(1, 48, 20, 24, generator=g), 2, "blend")            # N = 480: exact/linear blend window
    blk(torch.randn(1, 48, 24, 28, generator=g), 3, "linear")           # N = 672: random-feature attention
    blk(torch.randn(1, 48, 24, 28, generator=g), 4, "kvcap", regional_max_kv_tokens=64, shortcut=False)  # pooled-KV cap, no residual
    m = C2fMoA(64, 96, n=2, num_heads=6)
    case("c2f", m, lambda sd, xx, info: moa_ref.c2f_moa(sd, "m", xx, 6, info=info), torch.randn(2, 64, 16, 16, generator=g), 5)
