"""Golden vectors for `UltimateOptimizedMoE` (moe/modules.py:1534-1700), the MoE block of the v0_3 master YAMLs, from the REAL
reference on CPU, and the proof that oracle/ultimate_ref.py reproduces it bit for bit.

    python tests/golden/make_golden_v03.py         (build container only: needs /root/reference)

Writes tests/golden/v03_<case>.npz.  The whole v0_3 detector: tests/golden/make_golden_cfg5.py v03 (fwd_v03.npz)."""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import refboot, ultimate_ref  # noqa: E402
from tests.helpers import fill_by_name  # noqa: E402

refboot.boot()
from ultralytics.nn.modules.moe.modules import UltimateOptimizedMoE as Ref  # noqa: E402

CASES = {   # name: (ctor args, x shape, complexity bias)
    "base": ((128, 128, 4, 2, 0.5), (3, 128, 12, 16), 0.0),
    "e16": ((128, 128, 16, 2, 0.5), (4, 128, 8, 8), 0.0),
    "lowc": ((128, 128, 8, 2, 0.5), (2, 128, 10, 6), -4.0),      # complexity mean below 0.3: the lower clamp binds
    "k1": ((128, 128, 4, 1, 0.5), (3, 128, 9, 9), 0.0),
}

if __name__ == "__main__":
    torch.set_num_threads(4)
    for name, (args, xs, cbias) in CASES.items():
        m = Ref(*args).eval()
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.eps = 1e-3
        sd0 = m.state_dict()
        gen = {k: list(v.shape) for k, v in sd0.items() if v.is_floating_point() and v.dim() > 0}
        sd = {**{k: v.clone() for k, v in sd0.items() if k not in gen}, **fill_by_name(gen, seed=13, gain=1.0)}
        for k in sd:
            if k.endswith("routing.router.0.weight"):
                sd[k] = sd[k] * 2.0                              # spread the router scores without saturating the router's own Softmax
            if k.endswith("complexity_estimator.1.bias"):
                sd[k] = sd[k] + cbias
        m.load_state_dict(sd)
        x = torch.randn(*xs, generator=torch.Generator().manual_seed(3)) + torch.randn(xs[0], xs[1], 1, 1, generator=torch.Generator().manual_seed(4))
        with torch.inference_mode():
            y = m(x)
            info = {}
            oy = ultimate_ref.ultimate_optimized_moe({f"m.{k}": v for k, v in sd.items()}, "m", x, num_experts=args[2], top_k=args[3], split_ratio=args[4],
                                                     temperature=m.routing.temperature, info=info)
        exact = torch.equal(y, oy)
        r = info["m"]
        srt = r["probs"].log().sort(1, descending=True).values     # log-probabilities of the second softmax = its logits up to a constant
        gap = float((srt[:, args[3] - 1] - srt[:, args[3]]).min())
        print(f"[v03_{name}] oracle bit-exact vs reference: {exact}; T = {m.routing.temperature}; complexity {float(r['complexity']):.3f}; experts per image "
              f"{r['indices'].view(xs[0], -1).tolist()}; min logit gap at the cut {gap:.3e}")
        assert exact and gap > 2e-5   # 100x the fp32 evaluation-order noise of these logits (the router's softmax-of-a-softmax compresses them)
        rec = {"x": x.numpy(), "y": y.numpy(), "indices": r["indices"].view(xs[0], -1).numpy().astype(np.int32),
               "weights": r["weights"].view(xs[0], -1).numpy(), "keys": np.array(list(sd.keys())), "args": np.array(args)}
        for k, v in sd.items():
            rec[f"sd::{k}"] = v.numpy()
        np.savez_compressed(HERE / f"v03_{name}.npz", **rec)
    print("done")
