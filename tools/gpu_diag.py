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


def glds_ab(B=64, reps=20, extra=False):
    """Per-shape A/B of the opt-in tiled core (include/ymk_next.h) against what ymk_conv2d dispatches today, on the dense
    convolutions of YOLO-Master-S at batch B (bf16, SiLU): median of `reps` event-timed launches each; results compared."""
    import ctypes as C

    from yolo_master_amd import _lib

    lib = _lib.load()
    shapes = [  # Cin, Cout, k, stride, input H = W
        (128, 128, 3, 2, 160), (256, 256, 3, 2, 80), (128, 64, 3, 1, 80), (256, 512, 3, 2, 40), (256, 64, 3, 1, 40), (128, 128, 3, 2, 80),
        (256, 256, 3, 2, 40), (384, 256, 1, 1, 40), (256, 128, 1, 1, 40), (512, 128, 1, 1, 80), (768, 256, 1, 1, 40), (64, 64, 1, 1, 160),
        (128, 128, 3, 1, 40), (256, 64, 3, 1, 20)]
    if extra:   # the shapes the spatial-tile 3x3 kernel and the 128x128 tiled 1x1 kernel serve today
        shapes = [(64, 64, 3, 1, 80), (64, 64, 3, 1, 40), (64, 64, 3, 1, 20), (256, 768, 1, 1, 20), (768, 512, 1, 1, 20), (512, 256, 1, 1, 20),
                  (256, 256, 1, 1, 20), (384, 256, 1, 1, 20), (256, 128, 1, 1, 20), (128, 128, 1, 1, 20), (128, 64, 1, 1, 40), (64, 64, 1, 1, 80)]
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2] * 1e3

    print(f"{'shape':34s} {'today':>9s} {'glds s3':>9s} {'glds s2':>9s}   (us; TF/s of the best)")
    for cin, cout, k, s, hw in shapes:
        g = torch.Generator().manual_seed(cin + cout + k)
        x = torch.randn(B, hw, hw, cin, generator=g).to(torch.bfloat16).to(DEV)
        w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, generator=g) * (k * k * cin) ** -0.5, torch.bfloat16).to(DEV)
        bias = (torch.randn(cout, generator=g) * 0.1).to(DEV)
        ho = (hw + 2 * (k // 2) - k) // s + 1
        ys = [torch.empty((B, ho, ho, cout), dtype=torch.bfloat16, device=DEV) for _ in range(3)]
        d = _lib.ConvDesc(_lib.YMK_BF16, _lib.YMK_BF16, B, hw, hw, cin, cout, k, s, cin, cout, 0, w.shape[1], _lib.ACT_SILU)
        t0 = timed(lambda: lib.ymk_conv2d(C.byref(d), p(x), p(w), p(bias), None, p(ys[0]), st))
        var = lib.ymk_conv2d_last_variant()
        res = []
        for two, y in ((0, ys[1]), (1, ys[2])):
            if lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), two, st) != 0:
                res.append(float("nan"))
                continue
            res.append(timed(lambda: lib.ymk_conv2d_glds(C.byref(d), p(x), p(w), p(bias), None, p(y), two, st)))
            torch.cuda.synchronize()
            err = float((y.float() - ys[0].float()).abs().max())
            if err > 0.05 * max(1.0, float(ys[0].float().abs().max())):
                print(f"   MISMATCH two_stage={two}: max |d| {err:.3e}")
        best = min([t0] + [r for r in res if r == r])
        fl = 2.0 * B * ho * ho * cout * k * k * cin
        print(f"{cin:4d}->{cout:<4d} k{k} s{s} in {hw:3d}^2 (v{var})        {t0:9.1f} {res[0]:9.1f} {res[1]:9.1f}   {fl / best / 1e6:7.1f}")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "glds":
        glds_ab()
    if what == "glds2":
        glds_ab(extra=True)
        sys.exit(0)
    if what in ("all", "err"):
        errors()
    if what in ("all", "time"):
        layer_times()
    if what == "calls":
        call_times()
