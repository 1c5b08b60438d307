"""Pipelined serving step on one MI355X: what `predict()` does around the hot path, with the host link inside the timed region.

    host uint8 BGR frames (pinned) --H2D--> ymk_letterbox_preprocess -> forward -> batched NMS -> ymk_scale_boxes --D2H--> host (pinned)

Reference steps: `BasePredictor.preprocess` (LetterBox + BGR->RGB + /255, engine/predictor.py:155-178), `_predict_once`, `non_max_suppression`,
`scale_boxes` (models/yolo/detect/predict.py:109-122).  Four slots (device frame buffer + packed result buffer + pinned host result + compute
stream), one captured HIP graph per slot (letterbox -> ... -> scale_boxes), one H2D and one D2H stream: copies and the computes of
consecutive batches overlap (as bench.py --pipeline).  Reported next to the resident-input rate of bench.py (inputs already in HBM), which stays the headline value.

    python tools/serve_bench.py [--frames 720x1280] [--batch 64] [--steps 30] [--dtype bf16]     -> one JSON line (profiles/r03_serve.json)
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="720x1280", help="HxW of the host frames (uint8 BGR)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--scale", default="s")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--serial", action="store_true", help="no overlap: copy in, compute, copy out one after the other (A/B)")
    ap.add_argument("--slots", type=int, default=4, help="batches in flight (device frame buffer + packed result buffer + pinned host result + captured "
                    "graph + compute stream each).  Measured: 2 -> 6.4 k, 3 -> 8.9 k, 4 -> 10.4 k, 5 -> 10.4 k, 6 -> 11.1 k images/s (720p frames); with ONE "
                    "compute stream shared by two slots (the first version): 9.4-9.7 k")
    ap.add_argument("--split", type=int, default=1, help="sub-batches walked on parallel streams inside a slot's graph (as bench.py --split); "
                    "measured WORSE here: 2 -> 8.7-9.0 k images/s against 9.7 k (the copy engines' traffic shares the fabric with two compute streams)")
    a = ap.parse_args()

    from yolo_master_amd import ops, postprocess
    from yolo_master_amd._lib import check, lib
    from yolo_master_amd.nms import nms_padded
    from yolo_master_amd.nn.tasks import DetectionModel
    from yolo_master_amd.preprocess import letterbox_params
    from yolo_master_amd.weights import synth_state_dict

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    fh, fw = (int(v) for v in a.frames.split("x"))
    B, S, MAX_DET = a.batch, 640, 300
    model = DetectionModel(f"yolo-master-{a.scale}.yaml")
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model.eval().to(dev).set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype])

    g = torch.Generator().manual_seed(7)
    host_frames = [torch.randint(0, 256, (B, fh, fw, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)]
    p = letterbox_params((fh, fw), (S, S))
    nw, nh = p["new_unpad"]
    geom = torch.tensor([[fh, fw, nh, nw, p["top"], p["left"]]] * B, dtype=torch.int32, device=dev)
    offs = (torch.arange(B, dtype=torch.int64) * (fh * fw * 3)).to(dev)
    assert B % a.split == 0
    SUB = B // a.split
    sub_words = ops.nms_pack_numel(SUB, MAX_DET)
    words = a.split * sub_words          # per sub-batch: dets | idx | counts
    slots = []
    NS = max(2, a.slots)
    for _ in range(NS):
        slots.append({"stream": torch.cuda.Stream(device=dev), "frames": torch.empty((B, fh, fw, 3), dtype=torch.uint8, device=dev), "x": torch.empty((B, 3, S, S), dtype=torch.float32, device=dev),
                      "pack": torch.empty((words,), dtype=torch.float32, device=dev), "host": torch.empty((words,), dtype=torch.float32).pin_memory()})
    params = postprocess._params((S, S), [(fh, fw)] * B, None, dev)

    side = [torch.cuda.Stream(device=dev) for _ in range(a.split - 1)]

    def compute(sl):
        check(lib.ymk_letterbox_preprocess(ops._p(sl["frames"]), ops._p(offs), ops._p(geom), ops._p(sl["x"]), B, S, S, 114, 1, ops._stream()), "letterbox")

        def half(i):     # forward + NMS + box rescaling of sub-batch i into its slice of the packed result
            y, _ = model._predict_once(sl["x"][i * SUB:(i + 1) * SUB])
            dets, counts, _, _ = nms_padded(y, 0.25, 0.7, max_det=MAX_DET, pack=sl["pack"][i * sub_words:(i + 1) * sub_words])
            check(lib.ymk_scale_boxes(ops._p(dets), dets.stride(1), ops._p(counts), ops._p(params[i * SUB:(i + 1) * SUB]), SUB, MAX_DET, 1, 0,
                                      ops._stream()), "scale_boxes")

        cur = torch.cuda.current_stream()
        for st in side:
            st.wait_stream(cur)
        for i, st in enumerate(side, start=1):
            with torch.cuda.stream(st):
                half(i)
        half(0)
        for st in side:
            cur.wait_stream(st)

    s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    with torch.inference_mode():
        for sl in slots:
            with torch.cuda.stream(sl["stream"]):
                sl["frames"].copy_(host_frames[0], non_blocking=True)
                compute(sl)
            torch.cuda.synchronize()
        model.check_flags()
        for sl in slots:
            sl["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(sl["graph"], stream=sl["stream"]):
                compute(sl)
        torch.cuda.synchronize()

        def run(n):
            ev_in = [None] * NS       # H2D of the batch in this slot finished
            ev_cmp = [None] * NS      # compute of the batch in this slot finished (frames consumed, pack written)
            ev_out = [None] * NS      # D2H of this slot's pack finished (pack may be overwritten)
            for i in range(n):
                sl, k = slots[i % NS], i % NS
                s_cmp = sl["stream"]  # a compute stream per slot: consecutive batches overlap as in bench.py --pipeline
                with torch.cuda.stream(s_in):
                    if ev_cmp[k] is not None:
                        s_in.wait_event(ev_cmp[k])          # the previous batch in this slot has been read
                    sl["frames"].copy_(host_frames[i % 2], non_blocking=True)
                    ev_in[k] = s_in.record_event()
                with torch.cuda.stream(s_cmp):
                    s_cmp.wait_event(ev_in[k])
                    if ev_out[k] is not None:
                        s_cmp.wait_event(ev_out[k])         # the previous results of this slot are on the host
                    sl["graph"].replay()
                    ev_cmp[k] = s_cmp.record_event()
                with torch.cuda.stream(s_out):
                    s_out.wait_event(ev_cmp[k])
                    sl["host"].copy_(sl["pack"], non_blocking=True)
                    ev_out[k] = s_out.record_event()
                if a.serial:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()

        run(a.warmup)
        t0 = time.perf_counter()
        run(a.steps)
        dt = time.perf_counter() - t0
    dets, counts, _ = ops.nms_pack_views(slots[(a.steps - 1) % NS]["host"].view(a.split, sub_words), SUB, MAX_DET)
    h2d_mb, d2h_mb = B * fh * fw * 3 / 1e6, words * 4 / 1e6
    print(json.dumps({"metric": "images/sec, pipelined serving step (host uint8 frames -> letterbox -> forward -> NMS -> scale_boxes -> host)",
                      "value": round(B * a.steps / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / a.steps * 1e3, 4), "steps": a.steps,
                      "dtype": a.dtype, "overlap": not a.serial,
                      "config": {"workload": f"YOLO-Master-{a.scale.upper()}, {B} frames of {fh}x{fw}x3 uint8 per step -> 640x640", "h2d_mb_per_step": round(h2d_mb, 1),
                                 "d2h_mb_per_step": round(d2h_mb, 2), "launch": f"one hipGraph per slot, {NS} slots each with its compute stream, one H2D and one D2H stream"},
                      "detections_last_batch": int(counts.sum())}))


if __name__ == "__main__":
    main()
