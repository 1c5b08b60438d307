"""Golden vectors for the Segment head from the REAL reference: the v0 seg YAML
(`cfg/models/master/v0/seg/yolo-master-seg-n.yaml` = the detector with a Segment head) built by the reference's
SegmentationModel, seeded weights, CPU fp32.  Run in the build container:  python tests/golden/make_golden_seg.py
Writes tests/golden/fwd_seg_n.npz (+ keys_seg_n.json) and asserts the oracle (oracle/model_ref.py segment / proto) is
bit-exact against the reference."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import model_ref, refboot  # noqa: E402
from tests.helpers import fill_by_name  # noqa: E402

refboot.boot()
import yaml  # noqa: E402
from ultralytics.nn.tasks import SegmentationModel  # noqa: E402

REF_YAML = "/root/reference/ultralytics/cfg/models/master/v0/seg/yolo-master-seg-n.yaml"

if __name__ == "__main__":
    torch.set_num_threads(4)
    ref = SegmentationModel(REF_YAML, ch=3, nc=80, verbose=False)
    tmpl = ref.state_dict()
    spec = {k: list(v.shape) for k, v in tmpl.items() if v.is_floating_point() and v.dim() > 0 and not k.endswith("dfl.conv.weight")}
    sd = dict(tmpl)
    sd.update(fill_by_name(spec, seed=11, gain=0.8))
    ref.load_state_dict(sd)
    ref.eval()
    x = torch.rand(2, 3, 96, 64, generator=torch.Generator().manual_seed(4))
    taps = {}
    for m in ref.model:
        m.register_forward_hook(lambda mod, i, o, idx=m.i: taps.__setitem__(idx, o))
    with torch.inference_mode():
        out = ref(x)
    (y, proto), preds = out
    cfg = yaml.safe_load(open(REF_YAML))
    cfg["scale"] = "n"
    otaps = {}
    with torch.inference_mode():
        oy, _, _, omc, oproto = model_ref.forward(cfg, {k: v for k, v in ref.state_dict().items()}, x, fused=False, taps=otaps)
    exact = torch.equal(y, oy) and torch.equal(proto, oproto)
    print(f"[seg_n] y {tuple(y.shape)} proto {tuple(proto.shape)}; oracle bit-exact vs reference: {exact}; "
          f"max|dy| {(y - oy).abs().max().item():.3e} max|dproto| {(proto - oproto).abs().max().item():.3e}")
    assert exact
    rec = {"x": x.numpy(), "y": y.numpy(), "proto": proto.numpy(), "spec": json.dumps(spec), "cfg": json.dumps(cfg)}
    np.savez_compressed(HERE / "fwd_seg_n.npz", **rec)
    json.dump({k: list(v.shape) for k, v in tmpl.items()}, open(HERE / "keys_seg_n.json", "w"))
    print("wrote fwd_seg_n.npz", (HERE / "fwd_seg_n.npz").stat().st_size // 1024, "KB")
