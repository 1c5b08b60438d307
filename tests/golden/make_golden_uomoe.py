"""Golden vectors for `UltraOptimizedMoE` (moe/modules.py:121-232), the MoE block of v0_1/det/yolo-master-n-uomoe*.yaml and
exp/yolo-master-v0_2.yaml, from the REAL reference on CPU, and the proof that oracle/uomoe_ref.py reproduces it bit for bit.

    python tests/golden/make_golden_uomoe.py         (build container only: needs /root/reference)

Writes tests/golden/uomoe_<case>.npz.  The whole detector: tests/golden/make_golden_cfg5.py uomoe (fwd_uomoe.npz)."""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import refboot, uomoe_ref  # noqa: E402
from tests.helpers import fill_by_name  # noqa: E402

refboot.boot()
from ultralytics.nn.modules.moe.modules import UltraOptimizedMoE as Ref  # noqa: E402

CASES = {   # name: (ctor args, x shape, router gain)
    "base": ((64, 64, 4, 2), (3, 64, 24, 32), 8.0),          # the P3 row's shape: E = 4, top-2, map pooled 8x8 -> 3 x 4 router pixels
    "e16": ((128, 128, 16, 2), (4, 128, 16, 16), 8.0),       # 16 experts (the P5 row)
    "widen": ((64, 128, 8, 2), (2, 64, 20, 12), 8.0),        # in != out
    "small": ((64, 64, 4, 2), (2, 64, 8, 6), 8.0),           # map not larger than the router's 8x8 pool: no pre-pooling
    "thr": ((64, 64, 4, 2), (4, 64, 16, 16), -7.0),          # (negative: router gain 3, expert 1's bias raised by that much) a dominant expert: the second routed weight falls below 0.01 and is dropped
}

if __name__ == "__main__":
    torch.set_num_threads(4)
    for name, (args, xs, gain) in CASES.items():
        m = Ref(*args).eval()
        sd0 = m.state_dict()
        gen = {k: list(v.shape) for k, v in sd0.items() if v.is_floating_point() and v.dim() > 0}
        sd = {**{k: v.clone() for k, v in sd0.items() if k not in gen}, **fill_by_name(gen, seed=17, gain=1.0)}
        for k in sd:                                            # spread the router's logits: per-image choices differ
            if k.endswith("routing.router.6.weight"):
                sd[k] = sd[k] * (gain if gain > 0 else 3.0)
            if k.endswith("routing.router.6.bias") and gain < 0:
                sd[k][1] = sd[k][1] - gain
        m.load_state_dict(sd)
        x = torch.randn(*xs, generator=torch.Generator().manual_seed(3)) + torch.randn(xs[0], xs[1], 1, 1, generator=torch.Generator().manual_seed(4))
        with torch.inference_mode():
            y = m(x)
            info = {}
            oy = uomoe_ref.ultra_optimized_moe({f"m.{k}": v for k, v in sd.items()}, "m", x, top_k=args[3], info=info)
        exact = torch.equal(y, oy)
        r = info["m"]
        srt = r["probs"].sort(1, descending=True).values
        gap = float((srt[:, args[3] - 1] - srt[:, args[3]]).min()) if srt.shape[1] > args[3] else 1.0
        thr_gap = float((r["weights"] - 0.01).abs().min())
        dropped = int((r["weights"] <= 0.01).sum())
        print(f"[uomoe_{name}] oracle bit-exact vs reference: {exact}; experts per image {r['indices'].tolist()}; min pooled-weight gap at the cut "
              f"{gap:.3e}; routes dropped by the 0.01 threshold: {dropped} (closest weight to it: {thr_gap:.3e})")
        assert exact and gap > 1e-4 and thr_gap > 1e-4
        if name == "thr":
            assert dropped > 0, "the case must exercise the inference threshold"
        rec = {"x": x.numpy(), "y": y.numpy(), "indices": r["indices"].numpy().astype(np.int32), "weights": r["weights"].numpy(),
               "keys": np.array(list(sd.keys())), "args": np.array(args)}
        for k, v in sd.items():
            rec[f"sd::{k}"] = v.numpy()
        np.savez_compressed(HERE / f"uomoe_{name}.npz", **rec)
    print("done")
