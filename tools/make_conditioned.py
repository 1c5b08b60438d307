"""Generate yolo_master_amd/cfg/cond_<scale>.npz: overrides that turn the seeded random-weight detector
(yolo_master_amd/weights.py) into a WELL-CONDITIONED one for the parity fixtures at BASELINE's sizes (640 x 640).

Why.  A random-weight conv net with zero-mean BatchNorm outputs sits in the chaotic phase: every Conv+BN+SiLU
amplifies a relative perturbation (the BN removes the mean SiLU adds, so the Jacobian gain exceeds the signal gain),
and fp32 evaluation-order noise grows ~1.2x per layer — at 640 x 640 the reference's own fp32 result ends up
0.33 px / 1.5e-3 away from the exact (fp64) one with `bn_calib_*.npz`, which makes a 1e-4 parity bar meaningless
there.  Trained networks are not like that.  Three changes (all through the state_dict, the architecture is the
reference's):

  1. BatchNorm affine: weight ~ U(0.4, 0.6), bias ~ N(1.0, 0.3).  Pre-activations get a positive mean, SiLU works
     in its near-linear range, perturbations are no longer amplified (measured: per-layer reference-vs-fp64 noise
     stays at ~1e-5 through all 25 layers; class scores 3e-6).
  2. BatchNorm running statistics: calibrated on seeded synthetic 640 x 640 images (as tools/make_calibration.py).
  3. ES-MoE router output layer: logits centred over the calibration images and rescaled to a fixed spread, so that
     expert choice and the 0.4 importance threshold see both outcomes across a batch (without this the constant part
     of the pooled features picks the same expert for every image).

Steps 2-3 are iterated (router changes move the statistics of every later layer).  Runs the oracle restatement
(bit-exact vs the real reference) on CPU:    python tools/make_conditioned.py n s
"""
import sys
import zlib
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import model_ref  # noqa: E402
from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load  # noqa: E402
from yolo_master_amd.weights import synth_input, synth_state_dict  # noqa: E402

LOGIT_STD = 1.0
MOE_LAYERS = (3, 6, 9, 12)


def main(scales):
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    for scale in scales:
        cfg = yaml_model_load(f"yolo-master-{scale}.yaml")
        sd = synth_state_dict(DetectionModel(cfg).state_dict(), seed=0, calib=None)
        over = set()
        for k in sd:
            if ".bn." in k or ".norm.0." in k:
                g = torch.Generator().manual_seed(zlib.crc32(k.encode()) ^ 0x5EED)
                if k.endswith(".weight"):
                    sd[k] = torch.rand(sd[k].shape, generator=g) * 0.2 + 0.4
                elif k.endswith(".bias"):
                    sd[k] = torch.randn(sd[k].shape, generator=g) * 0.3 + 1.0
                if not k.endswith("num_batches_tracked"):
                    over.add(k)
        x = synth_input(16, 640, 640, seed=77)
        with torch.inference_mode():
            for it in range(4):
                model_ref.CALIBRATE = True
                try:
                    model_ref.forward(cfg, sd, x)
                finally:
                    model_ref.CALIBRATE = False
                info = {}
                model_ref.forward(cfg, sd, x, moe_info=info)
                for i in MOE_LAYERS:
                    lg = info[f"model.{i}"]["logits"]                     # [B, E]
                    mean = lg.mean(0)
                    s = LOGIT_STD / float((lg - mean).std().clamp_min(1e-6))
                    s = min(s, 1e4)
                    kw, kb = f"model.{i}.routing.routing_network.2.weight", f"model.{i}.routing.routing_network.2.bias"
                    sd[kw] = sd[kw] * s
                    sd[kb] = (sd[kb] - mean) * s
                    over.update((kw, kb))
                    print(f"[{scale}] iter {it} layer {i}: logit mean {[round(float(v), 2) for v in mean]} spread "
                          f"{float((lg - mean).std()):.3f} -> scale {s:.2f}; retained/expert "
                          f"{info[f'model.{i}']['retained'].sum(0).tolist()}")
        out = ROOT / "yolo_master_amd" / "cfg" / f"cond_{scale}.npz"
        np.savez_compressed(out, **{k: sd[k].numpy() for k in sorted(over)})
        print(out, len(over), "tensors", out.stat().st_size, "bytes")


if __name__ == "__main__":
    main(sys.argv[1:] or ["n", "s"])
