"""GPU box, with the reference staged (tools/stage_reference.sh): time the REAL reference beside libymk on the same MI355X and host.

  (1) the reference's own eager PyTorch-ROCm path: `DetectionModel(yaml).eval().fuse()` on cuda:0, fp32 and `.half()`, forward +
      its own `non_max_suppression` (conf 0.25, IoU 0.7), synchronised wall-clock per batch, p50 (benchmarks/suite.py:316-330) —
      the "what would a user get by just moving the reference to this GPU" baseline;
  (2) the same reference with libymk hooked underneath (`yolo_master_amd.enable`), same convention;
  (3) the reference on the GPU box's HOST cores (8 threads, fp32), forward only and forward + NMS — `cpu_baseline.kind = "reference"`
      measured where bench.py measures its port.

Writes gpurun_out/<tag>_reference_on_gpubox.json; the builder copies it to profiles/.

    python tools/gpu_reference_timing.py [scale=s] [batch=64] [tag=r04]
"""
import json
import os
import platform
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
if "YMK_REFERENCE" not in os.environ and (ROOT / ".refstage" / "ultralytics").is_dir():
    os.environ["YMK_REFERENCE"] = str(ROOT / ".refstage")

import torch  # noqa: E402

from oracle import refboot  # noqa: E402
from yolo_master_amd.weights import synth_input, synth_state_dict  # noqa: E402


def p50_sync(f, warm, n, cuda=True):
    ts = []
    for i in range(warm + n):
        if cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        if cuda:
            torch.cuda.synchronize()
        if i >= warm:
            ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


if __name__ == "__main__":
    scale = sys.argv[1] if len(sys.argv) > 1 else "s"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    tag = sys.argv[3] if len(sys.argv) > 3 else "r04"
    refboot.boot()
    refboot.stub_torchvision()
    from ultralytics.nn.tasks import DetectionModel as RefModel
    from ultralytics.utils.nms import non_max_suppression as ref_nms

    import yolo_master_amd

    yaml = f"{refboot.REF}/ultralytics/cfg/models/master/v0/det/yolo-master-{scale}.yaml"

    def make():
        m = RefModel(yaml, ch=3, nc=80, verbose=False)
        m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
        return m

    rec = {"model": f"YOLO-Master-{scale.upper()}", "batch": B, "imgsz": 640, "device": torch.cuda.get_device_name(0),
           "host": f"{platform.processor() or platform.machine()} ({os.cpu_count()} logical CPUs), torch {torch.__version__}",
           "convention": "synchronised wall-clock per batch, p50, images/s = batch * 1000 / p50_ms (benchmarks/suite.py:316-330); NMS conf 0.25 IoU 0.7"}
    x = synth_input(B, 640, 640, seed=1)
    with torch.inference_mode():
        # (1) un-hooked reference on the GPU
        for name, half in (("fp32", False), ("fp16", True)):
            m = make().eval().fuse(verbose=False).to("cuda:0")
            xin = x.to("cuda:0")
            if half:
                m, xin = m.half(), xin.half()
            fwd = p50_sync(lambda: m(xin), 3, 10)
            full = p50_sync(lambda: ref_nms(m(xin)[0].clone(), 0.25, 0.7), 2, 10)
            rec[f"reference_eager_gpu_{name}"] = {"forward_only_ms": round(fwd, 3), "forward_nms_ms": round(full, 3),
                                                  "forward_only_images_per_s": round(B * 1e3 / fwd, 1), "forward_nms_images_per_s": round(B * 1e3 / full, 1)}
            del m
            torch.cuda.empty_cache()
        # (2) the reference with libymk hooked underneath (fp32 input -> fp32 kernels; half input -> libymk_f16; bf16 via dtype=)
        import ultralytics.utils.nms as unms

        for name, dt, half in (("fp32", None, False), ("fp16", None, True), ("bf16", torch.bfloat16, False)):
            m = make()
            yolo_master_amd.enable(m, dtype=dt)
            m = m.eval().to("cuda:0")          # (the hook keeps its own packed weights; the reference module just moves)
            xin = x.to("cuda:0").half() if half else x.to("cuda:0")
            if half:
                m = m.half()
            fwd = p50_sync(lambda: m(xin), 3, 20)
            full = p50_sync(lambda: unms.non_max_suppression(m(xin)[0], 0.25, 0.7), 3, 20)
            rec[f"reference_hooked_libymk_{name}"] = {"forward_only_ms": round(fwd, 3), "forward_nms_ms": round(full, 3),
                                                       "forward_only_images_per_s": round(B * 1e3 / fwd, 1), "forward_nms_images_per_s": round(B * 1e3 / full, 1),
                                                       "hook_stats": yolo_master_amd.dropin.stats(m)}
            yolo_master_amd.disable(m)
            del m
            torch.cuda.empty_cache()
        # (3) the reference on this box's host cores (bounded sample: 4 images)
        cores = min(os.cpu_count() or 1, 8)
        torch.set_num_threads(cores)
        m = make().eval().fuse(verbose=False)
        xc = x[:4]
        fwd = p50_sync(lambda: m(xc), 1, 5, cuda=False)
        full = p50_sync(lambda: ref_nms(m(xc)[0].clone(), 0.25, 0.7), 1, 5, cuda=False)
        rec["reference_cpu_on_gpubox_host"] = {"cores": cores, "sample": "4x3x640x640 fp32, p50 of 5 passes", "forward_only_images_per_s": round(4e3 / fwd, 3),
                                               "forward_nms_images_per_s": round(4e3 / full, 3)}
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    json.dump(rec, open(out / f"{tag}_reference_on_gpubox.json", "w"), indent=1)
    print(json.dumps(rec))
