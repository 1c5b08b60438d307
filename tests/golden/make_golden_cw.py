"""Golden vectors for Cluster-Weighted NMS from the REFERENCE's own C++ implementation
(examples/YOLO-Master-Cross-Platform-Edge-Deployment/cpp/src/common.cpp:128-215), compiled in place into
oracle/_ref/libcwref.so by oracle/cwref/build.py.  Run in the build container (needs /root/reference and g++):

    python tests/golden/make_golden_cw.py

Writes tests/golden/cw_ref.npz: candidate sets and the reference's detections (standard and cluster-weighted mode).
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import nms_ref  # noqa: E402
from oracle.cwref import build as cwbuild  # noqa: E402


def load():
    lib = C.CDLL(str(cwbuild.build()))
    lib.cwref_nms_and_cap.restype = C.c_int
    lib.cwref_nms_and_cap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                      C.c_float, C.c_int, C.c_int, C.c_void_p]
    return lib


def ref_nms(lib, cands, conf, iou, max_det, cw, sigma, w, h):
    xywh = np.ascontiguousarray(np.stack([cands[:, 0], cands[:, 1], cands[:, 2] - cands[:, 0], cands[:, 3] - cands[:, 1]], 1), np.float32)
    sc = np.ascontiguousarray(cands[:, 4], np.float32)
    cl = np.ascontiguousarray(cands[:, 5], np.int32)
    out = np.zeros((max_det, 6), np.float32)
    n = lib.cwref_nms_and_cap(xywh.ctypes.data, sc.ctypes.data, cl.ctypes.data, len(cands), conf, iou, max_det, int(cw), sigma, w, h,
                              out.ctypes.data)
    return out[:n]


def make_cands(rng, n, nc, frame, clustered):
    if clustered:   # objects with many overlapping proposals around them (what CW-NMS is for)
        k = max(n // 12, 1)
        centres = rng.uniform(80, frame - 80, (k, 2))
        sizes = rng.uniform(30, 110, (k, 2))
        cls_k = rng.integers(0, nc, k)
        pick = rng.integers(0, k, n)
        xy = centres[pick] + rng.normal(0, 6, (n, 2))
        wh = sizes[pick] * rng.uniform(0.85, 1.15, (n, 2))
        cls = cls_k[pick]
    else:
        xy = rng.uniform(60, frame - 60, (n, 2))
        wh = rng.uniform(20, 100, (n, 2))
        cls = rng.integers(0, nc, n)
    score = rng.permutation(np.linspace(0.05, 0.98, n))   # distinct scores: std::sort is not stable
    b = np.concatenate([xy - wh / 2, xy + wh / 2], 1).clip(1, frame - 1)
    return np.concatenate([b, score[:, None], cls[:, None]], 1).astype(np.float32)


if __name__ == "__main__":
    lib = load()
    rng = np.random.default_rng(2024)
    rec = {}
    cases = [("clustered", 480, 3, 640, True, 0.25, 0.5, 0.1, 300), ("sparse", 300, 5, 640, False, 0.25, 0.45, 0.1, 300),
             ("capped", 600, 2, 512, True, 0.1, 0.6, 0.25, 40), ("big_pool", 3600, 1, 1024, True, 0.05, 0.5, 0.1, 300)]
    for name, n, nc, frame, clustered, conf, iou, sigma, max_det in cases:
        cands = make_cands(rng, n, nc, frame, clustered)
        std = ref_nms(lib, cands, conf, iou, max_det, False, sigma, frame, frame)
        cw = ref_nms(lib, cands, conf, iou, max_det, True, sigma, frame, frame)
        # cross-check the numpy restatement now (the CPU test repeats it without the library)
        f = cands[cands[:, 4] >= conf]
        keep = nms_ref.nms_greedy(f[:, :4] + f[:, 5:6] * 7680.0, f[:, 4], iou)[:max_det]
        mine = nms_ref.cw_refine(f, keep, iou, sigma)
        mine_xywh = np.stack([mine[:, 0], mine[:, 1], mine[:, 2] - mine[:, 0], mine[:, 3] - mine[:, 1]], 1)
        same_set = len(keep) == len(cw) and np.array_equal(f[keep, 4], cw[:, 4]) and np.array_equal(f[keep, 5], cw[:, 5])
        err = float(np.abs(mine_xywh - cw[:, :4]).max()) if same_set else float("nan")
        moved = float(np.abs(cw[:, :4] - std[:, :4]).max()) if len(cw) == len(std) else float("nan")
        print(f"[cw_{name}] n={n} kept {len(cw)} (standard {len(std)}); numpy oracle: same survivors {same_set}, max |box diff| {err:.3e} px; "
              f"cluster weighting moved boxes by up to {moved:.2f} px")
        assert same_set and err < 1e-3
        rec.update({f"{name}_cands": cands, f"{name}_std": std, f"{name}_cw": cw,
                    f"{name}_args": np.array([conf, iou, sigma, max_det, frame], np.float64)})
    np.savez_compressed(HERE / "cw_ref.npz", **rec)
    print("wrote", (HERE / "cw_ref.npz").stat().st_size, "bytes")
