"""Config 5 at model level (`v0_10/det/yolo-master-moa-mot-n.yaml`: VisualEnhancedAdaptiveGateMoE backbone, C2fMoT / C2fMoA
neck, Detect): the oracle's YAML walk with gated_ref / moa_ref / mot_ref against golden vectors produced by the REAL
reference model (tests/golden/make_golden_cfg5.py).  CPU only; the reference is not needed at run time."""
import json

import numpy as np
import torch

from tests.helpers import fill_by_name


def test_config5_model_oracle_reproduces_reference(golden_dir):
    from oracle import model_ref

    z = np.load(golden_dir / "fwd_cfg5.npz")
    cfg = json.loads(str(z["cfg"]))
    spec = json.loads(str(z["spec"]))
    sd = fill_by_name(spec, seed=5, gain=1.0)
    sd.update({k[len("fixed::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fixed::")})
    x = torch.from_numpy(z["x"])
    taps, info = {}, {}
    with torch.inference_mode():
        y, _, _ = model_ref.forward(cfg, sd, x, fused=False, taps=taps, moe_info=info)
    assert tuple(y.shape) == tuple(z["y_shape"])
    # discrete decisions first: routed experts of the gated blocks and top-k experts per token of every MoT block
    for k in [f for f in z.files if f.startswith("route::")]:
        assert np.array_equal(info[k[len("route::"):]]["indices"].numpy().astype(np.int16), z[k]), k
    # bit-identical at generation time (asserted by the generator); only the CPU thread count may differ here
    n = len(cfg["backbone"]) + len(cfg["head"])
    for i in range(n - 1):
        got = taps[i].reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        np.testing.assert_allclose(got, z[f"layer{i}_val"], rtol=1e-4, atol=1e-4, err_msg=f"layer {i} ({(cfg['backbone'] + cfg['head'])[i][2]})")
    got = y.reshape(-1)[torch.from_numpy(z["y_idx"].astype(np.int64))].numpy()
    np.testing.assert_allclose(got, z["y_val"], rtol=1e-4, atol=1e-3)
    kinds = {(cfg["backbone"] + cfg["head"])[i][2] for i in range(n)}
    assert {"VisualEnhancedAdaptiveGateMoE", "C2fMoA", "C2fMoT", "A2C2f", "C3k2", "Detect"} <= kinds
