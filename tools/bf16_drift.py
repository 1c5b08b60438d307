"""Mean |score_bf16 - score_fp32| of the N model on the fixture batch (the statistic of
tests/test_gpu_model.py::test_bf16_model_tracks_fp32), for A/B runs under YMK_DISABLE."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests.test_gpu_model import DEV, _model  # noqa: E402
from yolo_master_amd.weights import synth_input  # noqa: E402

for seed in (1, 2, 3):
    x = synth_input(4, 640, 640, seed=seed).to(DEV)
    with torch.inference_mode():
        m32 = _model("n")
        y32, _ = m32._predict_once(x)
        r32 = [(m32.model[i].last_route["gate_w"] > 0).cpu() for i in (3, 6, 9, 12)]
        m16 = _model("n", torch.bfloat16)
        y16, _ = m16._predict_once(x)
        r16 = [(m16.model[i].last_route["gate_w"] > 0).cpu() for i in (3, 6, 9, 12)]
    same = torch.stack([torch.stack([(a[b] == c[b]).all() for a, c in zip(r32, r16)]).all() for b in range(4)])
    d = (y16[:, 4:] - y32[:, 4:]).abs().mean(dim=(1, 2)).cpu()
    print(f"seed {seed}: same-routing {same.tolist()}  per-image mean|d| {[round(float(v), 4) for v in d]}")
