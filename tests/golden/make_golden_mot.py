"""Golden vectors for the Mixture-of-Transformer oracle (oracle/mot_ref.py), produced by the REAL reference modules
(`ultralytics.nn.modules.mot`) on CPU.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_mot.py

Writes tests/golden/mot_<case>.npz: the seeded state_dict, the input, the reference output, router weights and
top-k indices.  The script asserts that the oracle reproduces the reference bit for bit on every case.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
from oracle import mot_ref, refboot  # noqa: E402

refboot.boot()  # stubs cv2 and puts /root/reference on sys.path
from make_golden_moa import seeded_fill  # noqa: E402  (same deterministic parameter filler)

from ultralytics.nn.modules.mot.block import MoTBlock  # noqa: E402
from ultralytics.nn.modules.mot.wrappers import C2fMoT  # noqa: E402


def case(name, module, oracle_fn, x, seed, tweak=None):
    sd = seeded_fill(module, seed)
    if tweak:
        tweak(sd)
        module.load_state_dict(sd)
    module.eval()
    with torch.inference_mode():
        y = module(x)
        y = y[0] if isinstance(y, tuple) else y
        info = {}
        oy = oracle_fn({f"m.{k}": v for k, v in sd.items()}, x, info)
    exact = torch.equal(y, oy)
    used = [sorted(set(v["indices"].reshape(-1).tolist())) for v in info.values()]
    print(f"[mot_{name}] x {tuple(x.shape)} -> y {tuple(y.shape)}; oracle bit-exact vs reference: {exact}; "
          f"max|dy| {(y - oy).abs().max().item():.3e}; |y| max {y.abs().max().item():.3f}; experts used per block {used}")
    assert exact
    rec = {"x": x.numpy(), "y": y.numpy(), "keys": np.array(list(sd.keys())),
           "router_w": np.stack([v["weights"].numpy() for v in info.values()]),
           "router_idx": np.stack([v["indices"].numpy() for v in info.values()])}
    if any("scene_stats" in v for v in info.values()):
        rec["scene_stats"] = np.stack([v["scene_stats"].numpy() for v in info.values()])
        rec["scene_bias"] = np.stack([v["scene_bias"].numpy() for v in info.values()])
        ref_router = getattr(module, "router", None)
        if ref_router is not None and getattr(ref_router, "last_scene_stats", None) is not None:   # the reference's own record
            assert torch.equal(ref_router.last_scene_stats, next(iter(info.values()))["scene_stats"])
            assert torch.equal(ref_router.last_scene_bias, next(iter(info.values()))["scene_bias"])
    rec.update({f"sd::{k}": v.numpy() for k, v in sd.items()})
    np.savez_compressed(HERE / f"mot_{name}.npz", **rec)


if __name__ == "__main__":
    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(321)

    def blk(name, x, seed, tweak=None, **kw):
        m = MoTBlock(48, num_heads=6, **kw)
        okw = {k: v for k, v in kw.items() if k in ("top_k", "window_size", "n_points", "window_shift", "local_attn_window",
                                                    "use_spatial_router", "scene_aware_router", "scene_inference_mode")}
        case(name, m, lambda sd, xx, info: mot_ref.mot_block(sd, "m", xx, 6, info=info, **okw), x, seed, tweak)

    blk("top2", torch.randn(3, 48, 14, 18, generator=g), 1)                                  # padded windows, top-2 of 3
    blk("shift", torch.randn(2, 48, 16, 20, generator=g), 2, window_shift=True, local_attn_window=7)  # shifted windows, local windows
    blk("top1", torch.randn(2, 48, 12, 12, generator=g), 3, top_k=1)
    blk("dense", torch.randn(1, 48, 10, 14, generator=g), 4, top_k=3)                         # every expert everywhere

    def never_deformable(sd):   # expert 2 is never selected: the per-sample dispatch must skip it entirely
        sd["router.router.3.bias"] = torch.tensor([0.3, -0.2, -50.0])
    blk("skip", torch.randn(2, 48, 9, 11, generator=g), 5, tweak=never_deformable)
    # round 4: the scene-aware residual (mot/router.py:166-240; map sizes that the 2x2 / 4x4 adaptive pools do not divide), the
    # image-level router, both together, and the "bypass" inference policy
    if len(sys.argv) > 1 and sys.argv[1] == "scene":
        blk("scene", torch.randn(3, 48, 14, 18, generator=g) * torch.tensor([0.5, 1.0, 2.0]).view(3, 1, 1, 1), 11, scene_aware_router=True)
        blk("scene3", torch.randn(2, 48, 3, 9, generator=g), 12, scene_aware_router=True, scene_hidden_dim=5, top_k=1)
        blk("image", torch.randn(3, 48, 12, 16, generator=g), 13, use_spatial_router=False)
        blk("image_scene", torch.randn(2, 48, 10, 10, generator=g), 14, use_spatial_router=False, scene_aware_router=True)
        blk("scene_bypass", torch.randn(2, 48, 8, 12, generator=g), 15, scene_aware_router=True, scene_inference_mode="bypass")
        sys.exit(0)
    m = C2fMoT(64, 96, n=2, num_heads=6)
    case("c2f", m, lambda sd, xx, info: mot_ref.c2f_mot(sd, "m", xx, 6, info=info), torch.randn(2, 64, 15, 17, generator=g), 6)
