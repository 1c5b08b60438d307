"""`ymk_letterbox_preprocess` (csrc/preproc.hip) on the CPU lane emulator against oracle/pre_ref.py, BIT-EXACT (integer resize
arithmetic, IEEE /255): up- and down-scaling, the exact-2x case, no-resize, odd sizes, the reference's two asset sizes
(bus.jpg 1080x810, zidane.jpg 720x1280), mixed shapes in one batch.  The same cases run on the GPU (tests/test_gpu_next.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pre_ref

CASES = [
    # list of (h, w) per batch, target (H, W)
    ([(48, 64)], (64, 64)),
    ([(64, 64), (31, 64), (64, 17)], (64, 64)),              # no resize / pad rows / pad columns
    ([(128, 128), (256, 256)], (64, 64)),                    # exact 2x downscale and 4x downscale
    ([(37, 53), (90, 41), (7, 5)], (96, 96)),                # upscaling, odd sizes
    ([(1080, 810), (720, 1280)], (640, 640)),                # the reference's asset sizes
    ([(333, 500)], (384, 672)),                              # non-square target
]


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def run_case(lib, case, dev="cpu", stream=None):
    from yolo_master_amd.preprocess import letterbox_params

    shapes, (H, W) = case
    rng = np.random.default_rng(len(shapes) * 1000 + H)
    imgs = []
    for (h, w) in shapes:   # smooth structure + noise, so that interpolation errors are not hidden by randomness alone
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.stack([(xx * 255) // max(w - 1, 1), (yy * 255) // max(h - 1, 1), ((xx + yy) * 7) % 256], -1)).astype(np.int32)
        imgs.append(np.clip(base + rng.integers(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8))
    want = pre_ref.preprocess(imgs, (H, W))
    geoms, offs, off = [], [], 0
    for im in imgs:
        h, w = im.shape[:2]
        p = letterbox_params((h, w), (H, W))
        geoms.append([h, w, p["new_unpad"][1], p["new_unpad"][0], p["top"], p["left"]])
        offs.append(off)
        off += h * w * 3
    src = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).to(dev)
    geom = torch.tensor(geoms, dtype=torch.int32, device=dev)
    offd = torch.tensor(offs, dtype=torch.int64, device=dev)
    out = torch.full((len(imgs), 3, H, W), -1.0, dtype=torch.float32, device=dev)
    assert lib.ymk_letterbox_preprocess(_p(src), _p(offd), _p(geom), _p(out), len(imgs), H, W, 114, 1, stream) == 0
    got = out.cpu().numpy()
    if not np.array_equal(got, want):
        d = np.abs(got - want)
        raise AssertionError(f"{case}: {int((d > 0).sum())} of {d.size} values differ, max {d.max() * 255:.3f} grey levels")


@pytest.mark.parametrize("case", CASES[:4] + CASES[5:])
def test_letterbox_kernel_on_emulator(case, hostlib):
    run_case(hostlib, case)
