"""Time the REAL reference (imported from /root/reference, BASELINE.md section 2's protocol: `DetectionModel(yaml).eval().fuse()`,
`torch.inference_mode()`, fp32, forward + `non_max_suppression(conf 0.25, IoU 0.7)`, 8 threads) beside the oracle port that
bench.py's `cpu_baseline` times on the GPU box, on the SAME host, weights and images.  Build container only (the GPU box has no
reference checkout — unless tools/stage_reference.sh put one beside the snapshot for that call: YMK_REFERENCE); writes
profiles/<tag>_cpu_reference.json, which bench.py quotes next to its own `cpu_baseline`.  Forward only and forward + NMS are timed
separately: the reference's Python NMS dominates once thousands of candidates pass the threshold.

    python tools/cpu_reference_timing.py [scale=s] [batch=4] [passes=5] [tag=r04]
"""
import json
import os
import platform
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import model_ref, nms_ref, refboot  # noqa: E402
from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load  # noqa: E402
from yolo_master_amd.weights import synth_input, synth_state_dict  # noqa: E402

refboot.boot()
from ultralytics.nn.tasks import DetectionModel as RefModel  # noqa: E402
from ultralytics.utils.nms import non_max_suppression as ref_nms  # noqa: E402


def p50(f, n):
    f()
    ts = []
    for _ in range(n):
        t0 = time.time()
        f()
        ts.append(time.time() - t0)
    ts.sort()
    return ts[len(ts) // 2]


if __name__ == "__main__":
    scale = sys.argv[1] if len(sys.argv) > 1 else "s"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    tag = sys.argv[4] if len(sys.argv) > 4 else "r04"
    cores = min(os.cpu_count() or 1, 8)
    torch.set_num_threads(cores)
    cfg = yaml_model_load(f"yolo-master-{scale}.yaml")
    sd = synth_state_dict(DetectionModel(cfg).state_dict(), seed=0)
    ref = RefModel(f"{refboot.REF}/ultralytics/cfg/models/master/v0/det/yolo-master-{scale}.yaml", ch=3, nc=80, verbose=False)
    ref.load_state_dict(sd)
    ref.eval().fuse(verbose=False)
    x = synth_input(B, 640, 640, seed=1)
    out = {}
    with torch.inference_mode():
        def run_ref():
            y = ref(x)
            out["ref"] = ref_nms((y[0] if isinstance(y, (tuple, list)) else y).clone(), 0.25, 0.7)

        def run_port():
            y, _, _ = model_ref.forward(cfg, sd, x)
            out["port"] = nms_ref.non_max_suppression(y.numpy(), 0.25, 0.7)

        t_ref, t_port = p50(run_ref, n), p50(run_port, n)
        t_ref_fwd, t_port_fwd = p50(lambda: ref(x), n), p50(lambda: model_ref.forward(cfg, sd, x), n)
    same = all(len(a) == len(b) for a, b in zip(out["ref"], out["port"]))
    rec = {"reference_images_per_s": round(B / t_ref, 3), "port_images_per_s": round(B / t_port, 3),
           "reference_forward_only_images_per_s": round(B / t_ref_fwd, 3), "port_forward_only_images_per_s": round(B / t_port_fwd, 3), "cores": cores,
           "host": f"{platform.processor() or platform.machine()} ({os.cpu_count()} logical CPUs), torch {torch.__version__}",
           "sample": f"YOLO-Master-{scale.upper()} fp32 forward + NMS, {B}x3x640x640, p50 of {n} passes; kind 'reference' = ultralytics from /root/reference "
                     f"(fused, inference_mode), kind 'port' = oracle/model_ref + nms_ref (what bench.py times on the GPU box)",
           "same_detection_counts": same}
    (ROOT / "profiles").mkdir(exist_ok=True)
    json.dump(rec, open(ROOT / "profiles" / f"{tag}_cpu_reference.json", "w"), indent=1)
    print(json.dumps(rec))
