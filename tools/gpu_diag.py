"""GPU diagnostics: per-layer error vs golden (fp32, N) and per-layer time (bf16, S, bs=64)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from yolo_master_amd import ops  # noqa: E402
from yolo_master_amd.nn.tasks import DetectionModel  # noqa: E402
from yolo_master_amd.weights import synth_input, synth_state_dict  # noqa: E402

DEV = "cuda:0"


def build(scale, dtype):
    m = DetectionModel(f"yolo-master-{scale}.yaml")
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
    return m.eval().to(DEV).set_compute_dtype(dtype)


def errors(case="n640"):
    z = np.load(ROOT / "tests" / "golden" / f"fwd_{case}.npz")
    m = build(chr(int(z["scale"])), torch.float32)
    x = synth_input(int(z["B"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]))
    taps = {}
    with torch.inference_mode():
        y, _ = m._predict_once(x.to(DEV), taps=taps)
    for i in range(25):
        t = taps[i]
        if not torch.is_tensor(t):
            t = t.materialise()
        got = ops.nhwc_to_nchw_f32(t).cpu().reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].numpy()
        ref = z[f"layer{i}_val"]
        e = np.abs(got - ref)
        print(f"layer {i:2d} {m.model[i].type:12s} max|ref| {np.abs(ref).max():8.3f}  max err {e.max():.3e}  "
              f"rel-to-max {e.max() / np.abs(ref).max():.2e}  mean err {e.mean():.2e}")
    got = y.cpu().reshape(-1)[torch.from_numpy(z["y_idx"].astype(np.int64))].numpy()
    e = np.abs(got - z["y_val"])
    print(f"y: max err {e.max():.3e} (max|ref| {np.abs(z['y_val']).max():.1f}); rel err max {np.max(e / (np.abs(z['y_val']) + 1e-3)):.2e}")


def layer_times(scale="s", B=64, dtype=torch.bfloat16, reps=5):
    m = build(scale, dtype)
    x = synth_input(B, 640, 640, seed=1).to(DEV)
    from yolo_master_amd.nms import nms_padded

    with torch.inference_mode():
        for _ in range(2):
            y, _ = m._predict_once(x)
        torch.cuda.synchronize()
        # instrument: time each top-level layer by re-walking the graph with events
        import torch.nn as nn
        from yolo_master_amd.nn.modules import Concat, Conv, Detect, ES_MOE, LazyUpsample

        tot = {}
        for rep in range(reps):
            ys, cur = [], x
            for mod in m.model:
                if mod.f != -1:
                    cur = ys[mod.f] if isinstance(mod.f, int) else [cur if j == -1 else ys[j] for j in mod.f]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if isinstance(mod, Conv):
                    cur = mod._run_stem(cur) if mod.i == 0 else mod._run(cur)
                elif isinstance(mod, nn.Upsample):
                    cur = LazyUpsample(cur)
                elif isinstance(mod, Concat):
                    cur = mod._run(cur)
                elif isinstance(mod, Detect):
                    cur, raw = mod._run(cur)
                else:
                    cur = mod._run(cur)
                e1.record()
                ys.append(cur if mod.i in m.save else None)
                tot.setdefault(mod.i, []).append((e0, e1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nms_padded(cur, 0.25, 0.7)
            e1.record()
            tot.setdefault(99, []).append((e0, e1))
        torch.cuda.synchronize()
        s = 0.0
        for i, evs in tot.items():
            ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
            s += ms
            name = "NMS" if i == 99 else m.model[i].type
            print(f"layer {i:2d} {name:12s} {ms:8.3f} ms")
        print(f"sum {s:.3f} ms (eager, includes launch gaps) -> {B / s * 1e3:.0f} img/s")


def call_times(scale="s", B=64, dtype=torch.bfloat16, reps=3):
    """Every op call of one forward+NMS with its shape, time, algorithmic GB/s and TFLOP/s (ops.TIMER)."""
    from yolo_master_amd.nms import nms_padded

    m = build(scale, dtype)
    x = synth_input(B, 640, 640, seed=1).to(DEV)
    with torch.inference_mode():
        for _ in range(2):
            nms_padded(m(x)[0], 0.25, 0.7)
        torch.cuda.synchronize()
        ops.TIMER.start()
        for _ in range(reps):
            nms_padded(m(x)[0], 0.25, 0.7)
        torch.cuda.synchronize()
        ops.TIMER.stop()
    n = len(ops.TIMER.records) // reps
    rows = []
    for i in range(n):
        fam, _, _, nb, fl = ops.TIMER.records[i]
        ms = sorted(ops.TIMER.records[i + r * n][1].elapsed_time(ops.TIMER.records[i + r * n][2]) for r in range(reps))[reps // 2]
        rows.append((ms, i, fam, ops.TIMER.shapes[i], nb, fl))
    print(f"{n} op calls, {sum(r[0] for r in rows):.3f} ms")
    for ms, i, fam, shp, nb, fl in sorted(rows, reverse=True):
        print(f"{i:3d} {fam[:58]:58s} {shp:34s} {ms * 1e3:8.1f} us {nb / ms / 1e6:8.0f} GB/s {fl / ms / 1e9:8.1f} TF/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "err"):
        errors()
    if what in ("all", "time"):
        layer_times()
    if what == "calls":
        call_times()
