"""Generate yolo_master_amd/cfg/bn_calib_<scale>.npz: BatchNorm running statistics that normalise the
seeded random-weight network (yolo_master_amd/weights.py) on seeded synthetic images.

Random conv weights + random BN statistics give a chaotic network whose activations explode or whose
outputs are spatially constant (score ties, identical routing for every image).  Real checkpoints are
normalised by training; with no checkpoint available offline we emulate that once, on CPU, with the
oracle's forward restatement in CALIBRATE mode, and commit the resulting statistics (a few hundred KB)
so that every machine (this container, the GPU box) builds the identical state_dict.

    python tools/make_calibration.py n s
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import model_ref  # noqa: E402
from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load  # noqa: E402
from yolo_master_amd.weights import synth_input, synth_state_dict  # noqa: E402


def main(scales):
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    for scale in scales:
        cfg = yaml_model_load(f"yolo-master-{scale}.yaml")
        sd = synth_state_dict(DetectionModel(cfg).state_dict(), seed=0, calib=None)
        x = synth_input(16, 640, 640, seed=77)
        with torch.inference_mode():
            model_ref.CALIBRATE = True
            try:
                model_ref.forward(cfg, sd, x)
            finally:
                model_ref.CALIBRATE = False
        stats = {k: v.numpy() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
        out = ROOT / "yolo_master_amd" / "cfg" / f"bn_calib_{scale}.npz"
        np.savez_compressed(out, **stats)
        print(out, len(stats), "tensors", out.stat().st_size, "bytes")


if __name__ == "__main__":
    main(sys.argv[1:] or ["n", "s"])
