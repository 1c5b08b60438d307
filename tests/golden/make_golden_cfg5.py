"""Model-level golden vectors for config 5 (`v0_10/det/yolo-master-moa-mot-n.yaml`: gated-MoE backbone + MoA/MoT
neck), produced by the REAL reference model on CPU, and the proof that the oracle (oracle/model_ref.forward with
gated_ref / moa_ref / mot_ref) reproduces it bit for bit.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_cfg5.py

The state_dict is not stored (7 M parameters): tests/helpers.fill_by_name regenerates it from the committed
name -> shape spec; fixed buffers (random-feature bases, temperatures, counters) are stored as they are.
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch
import yaml

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import model_ref, refboot  # noqa: E402
from tests.helpers import fill_by_name  # noqa: E402  (before boot(): the reference checkout has a `tests` package too)

refboot.boot()
from ultralytics.nn.tasks import DetectionModel as RefModel  # noqa: E402

YAML = Path(refboot.REF) / "ultralytics/cfg/models/master/v0_10/det/yolo-master-moa-mot-n.yaml"
NS = 256
TAG = "cfg5"
if len(sys.argv) > 1 and sys.argv[1] == "v15":   # the v0_15 generation: GatedFusionMoE backbone on the v0 neck / head
    YAML = Path(refboot.REF) / "ultralytics/cfg/models/master/v0_15/det/yolo-master-n.yaml"
    TAG = "v15"
if len(sys.argv) > 1 and sys.argv[1] in ("v01", "v03", "v04", "v05", "v06", "v07", "v08", "v09"):   # earlier generations (one MoE class each): v0_1 = ModularRouterExpertMoE, v0_3 = UltimateOptimizedMoE, v0_4 ... = the gated family
    TAG = sys.argv[1]
    YAML = Path(refboot.REF) / f"ultralytics/cfg/models/master/v0_{int(TAG[1:])}/det/yolo-master-n.yaml"


if len(sys.argv) > 1 and sys.argv[1] == "v08s":   # v0_8 with SharedExpertMoE (moe/shared_expert_moe.py): the P3 and P4 blocks share one expert pool
    TAG = "v08s"
    YAML = Path(refboot.REF) / "ultralytics/cfg/models/master/v0_8/det/yolo-master-moe-mot-shared-n.yaml"


if len(sys.argv) > 1 and sys.argv[1] == "uomoe":   # v0_1 with UltraOptimizedMoE blocks (moe/modules.py:121-232)
    TAG = "uomoe"
    YAML = Path(refboot.REF) / "ultralytics/cfg/models/master/v0_1/det/yolo-master-n-uomoe.yaml"


def sample_idx(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(n, generator=g)[: min(k, n)].sort().values


if __name__ == "__main__":
    torch.set_num_threads(4)
    ref = RefModel(str(YAML), ch=3, nc=80, verbose=False)
    sd0 = ref.state_dict()
    gen = {k: list(v.shape) for k, v in sd0.items() if v.is_floating_point() and v.dim() > 0 and not k.endswith("_rf_matrix")}
    fixed = {k: v.clone() for k, v in sd0.items() if k not in gen}
    sd = {**fill_by_name(gen, seed=5, gain=1.0), **fixed}
    ref.load_state_dict(sd)
    ref.eval()
    g = torch.Generator().manual_seed(55)
    x = torch.rand(2, 3, 192, 160, generator=g) if TAG == "cfg5" else torch.rand(3, 3, 128, 96, generator=g)
    taps = {}
    for m in ref.model:
        m.register_forward_hook(lambda mod, i, o, idx=m.i: taps.__setitem__(idx, o))
    with torch.inference_mode():
        y = ref(x)
        y = y[0] if isinstance(y, (tuple, list)) else y
    cfg = yaml.safe_load(open(YAML))
    # width / depth scaling of the YAML rows is resolved by the reference parser; the oracle only needs the module
    # names, the from-indices and the non-channel arguments, which scaling leaves untouched
    otaps, info = {}, {}
    with torch.inference_mode():
        oy, _, _ = model_ref.forward(cfg, sd, x, fused=False, taps=otaps, moe_info=info)
    n = len(ref.model)
    worst = max((float((taps[i] - otaps[i]).abs().max()) if torch.is_tensor(taps[i]) else 0.0) for i in range(n - 1))
    exact = all(torch.equal(taps[i], otaps[i]) for i in range(n - 1)) and torch.equal(y, oy)
    mags = [round(float(taps[i].abs().max()), 2) for i in range(n - 1)]
    print(f"[{TAG}] {n} layers, {sum(v.numel() for v in sd.values()) / 1e6:.2f} M values; oracle bit-exact vs reference: {exact}; "
          f"worst layer |d| {worst:.3e}; max|dy| {(y - oy).abs().max().item():.3e}")
    print(f"[{TAG}] per-layer max |activation|:", mags)
    assert exact
    rec = {"x": x.numpy(), "spec": np.array(json.dumps(gen)), "cfg": np.array(json.dumps({"scale": "n", **{k: cfg[k] for k in ("nc", "scales", "backbone", "head")}})),
           "y_shape": np.array(y.shape)}
    for k, v in fixed.items():
        rec[f"fixed::{k}"] = v.numpy()
    for i in range(n - 1):
        idx = sample_idx(taps[i].numel(), NS, 500 + i)
        rec[f"layer{i}_idx"], rec[f"layer{i}_val"] = idx.numpy().astype(np.int32), taps[i].reshape(-1)[idx].numpy()
    idx = sample_idx(y.numel(), 4096, 9)
    rec["y_idx"], rec["y_val"] = idx.numpy().astype(np.int32), y.reshape(-1)[idx].numpy()
    for k, v in info.items():
        if "indices" in v:
            rec[f"route::{k}"] = v["indices"].numpy().astype(np.int16)
    np.savez_compressed(HERE / f"fwd_{TAG}.npz", **rec)
    json.dump({k: list(v.shape) for k, v in sd0.items()}, open(HERE / f"keys_{TAG}.json", "w"))   # ordered: the drop-in contract
    print(f"[{TAG}] wrote", (HERE / f"fwd_{TAG}.npz").stat().st_size, "bytes")
