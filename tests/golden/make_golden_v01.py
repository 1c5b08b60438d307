"""Golden vectors for `ModularRouterExpertMoE` (= OptimizedMOEImproved, moe/modules.py:957-1198), the MoE block of the v0_1 master
YAMLs, from the REAL reference on CPU, and the proof that oracle/modular_ref.py reproduces it bit for bit.

    python tests/golden/make_golden_v01.py         (build container only: needs /root/reference)

Writes tests/golden/v01_<case>.npz: x, the module's state_dict, y, routing weights / indices.  The whole v0_1 detector is
covered by tests/golden/make_golden_cfg5.py v01 (fwd_v01.npz)."""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import modular_ref, refboot  # noqa: E402
from tests.helpers import fill_by_name  # noqa: E402

refboot.boot()
from ultralytics.nn.modules.moe.modules import ModularRouterExpertMoE as Ref  # noqa: E402

CASES = {   # name: (ctor args, kwargs, x shape)
    "base": ((64, 64, 4, 2), {}, (3, 64, 12, 16)),          # v0_1 row shape: E = 4, top-2, residual
    "e16": ((128, 128, 16, 2), {}, (4, 128, 8, 8)),          # 16 experts (the P5 row of the YAML)
    "widen": ((64, 128, 8, 2), {}, (2, 64, 10, 6)),          # in != out: no residual
    "small": ((64, 64, 4, 2), {}, (2, 64, 4, 3)),            # map not larger than the router's 4x4 pool: no pre-pooling
    "k1": ((64, 64, 4, 1), {}, (3, 64, 9, 9)),               # top-1
}

if __name__ == "__main__":
    torch.set_num_threads(4)
    for name, (args, kw, xs) in CASES.items():
        m = Ref(*args, **kw).eval()
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.eps = 1e-3                                  # what DetectionModel's initialize_weights sets on every BatchNorm2d
        sd0 = m.state_dict()
        gen = {k: list(v.shape) for k, v in sd0.items() if v.is_floating_point() and v.dim() > 0}
        sd = {**{k: v.clone() for k, v in sd0.items() if k not in gen}, **fill_by_name(gen, seed=11, gain=1.0)}
        for k in sd:                                            # spread the router's logits: per-image choices differ
            if k.endswith("routing.router.3.weight"):
                sd[k] = sd[k] * 6.0
        m.load_state_dict(sd)
        x = torch.randn(*xs, generator=torch.Generator().manual_seed(3)) + torch.randn(xs[0], xs[1], 1, 1, generator=torch.Generator().manual_seed(4))
        with torch.inference_mode():
            y = m(x)
            info = {}
            oy = modular_ref.modular_router_expert_moe({f"m.{k}": v for k, v in sd.items()}, "m", x, top_k=args[3], info=info)
        exact = torch.equal(y, oy)
        idx = info["m"]["indices"]
        probs = info["m"]["probs"]
        srt = probs.sort(1, descending=True).values
        gap = float((srt[:, args[3] - 1] - srt[:, args[3]]).min()) if srt.shape[1] > args[3] else 1.0
        print(f"[v01_{name}] oracle bit-exact vs reference: {exact}; experts per image {idx.tolist()}; min prob gap at the cut {gap:.3e}")
        assert exact and gap > 1e-3
        rec = {"x": x.numpy(), "y": y.numpy(), "indices": idx.numpy().astype(np.int32), "weights": info["m"]["weights"].numpy(),
               "keys": np.array(list(sd.keys())), "args": np.array(args)}
        for k, v in sd.items():
            rec[f"sd::{k}"] = v.numpy()
        np.savez_compressed(HERE / f"v01_{name}.npz", **rec)
    print("done")
