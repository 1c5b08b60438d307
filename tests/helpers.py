"""Shared helpers for the parity tests (HIP path vs oracle / golden)."""
from __future__ import annotations

import numpy as np
import torch

from yolo_master_amd.weights import synth_state_dict

# tolerances (stated once, used by every parity test)
#   fp32: the HIP path accumulates in fp32 with a different summation order than oneDNN; per-op relative
#         error ~1e-6, end-to-end (26 layers, random calibrated weights) we require |d| <= 1e-4 + 1e-4*|ref|
#         on boxes/scores (north_star: 1e-4 fp32).
#   bf16: activations and weights rounded to bf16 (8 mantissa bits) at every layer; per-op bound 2e-2 rel.
#   fp16: the fp16 build (libymk_f16.so): 10 mantissa bits, per-op bound 4e-3 rel.
TOL = {torch.float32: dict(rtol=1e-4, atol=1e-4), torch.bfloat16: dict(rtol=3e-2, atol=3e-2), torch.float16: dict(rtol=4e-3, atol=4e-3)}


def nhwc(x_nchw: torch.Tensor, dtype, dev, pad_c: int = 0, c_off: int = 0) -> torch.Tensor:
    """NCHW CPU tensor -> NHWC device view; with pad_c>0 the view is a channel slice of a wider buffer."""
    B, C, H, W = x_nchw.shape
    buf = torch.full((B, H, W, C + pad_c), 7.0, dtype=dtype, device=dev)
    v = buf[..., c_off:c_off + C]
    v.copy_(x_nchw.permute(0, 2, 3, 1).to(dtype))
    return v


def nchw(y_nhwc: torch.Tensor) -> torch.Tensor:
    return y_nhwc.float().permute(0, 3, 1, 2).cpu()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


def assert_close(got: torch.Tensor, ref: torch.Tensor, dtype, what: str, scale_aware=True):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} != {tuple(ref.shape)}"
    tol = TOL[dtype]
    mag = ref.abs().max().item() if scale_aware else 1.0
    err = (got - ref).abs()
    bound = tol["atol"] * max(mag, 1.0) + tol["rtol"] * ref.abs()
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max err "
                           f"{err.max().item():.3e} (ref max {mag:.3e}), first bad idx {bad.nonzero()[0].tolist()}")
    return err.max().item()


def module_sd(module, prefix="model.0", seed=0):
    """Seeded random parameters for a module, returned both loaded into it and as a prefixed oracle dict."""
    sd = synth_state_dict(module.state_dict(), seed=seed, calib=None)
    module.load_state_dict(sd)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3
    return {f"{prefix}.{k}": v for k, v in sd.items()}


def load_npz(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def fill_by_name(spec: dict, seed: int = 0, gain: float = 1.0) -> dict:
    """Deterministic parameters from names and shapes alone (no module needed): every tensor draws from its own
    generator seeded by crc32(name) ^ seed.  Used for fixtures of models whose state_dict is too large to commit:
    the fixture stores the spec (name -> shape) and both sides regenerate the same values.  Integer tensors, scalars
    and fixed bases (`_rf_matrix`) are not generated — the caller supplies them."""
    import zlib

    out = {}
    for k, shape in spec.items():
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) ^ seed) & 0x7FFFFFFF)
        shape = tuple(shape)
        if k.endswith("running_var"):
            out[k] = 0.5 + torch.rand(shape, generator=g)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(shape, generator=g)
        elif k.endswith(".bias") or k.endswith("expert_norm_bias"):
            out[k] = 0.1 * torch.randn(shape, generator=g)
        elif any(t in k for t in ("ls_attn", "ls_ffn", ".ls1", ".ls2")):
            out[k] = 0.3 + 0.05 * torch.randn(shape, generator=g)
        elif k.endswith(("_scale", ".alpha", ".gamma")):
            out[k] = 0.3 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1 or k.endswith("expert_norm_weight"):   # norm scales
            out[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:                                                        # conv / linear weights: fan-in scaling
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            out[k] = torch.randn(shape, generator=g) * (gain / max(fan_in, 1) ** 0.5)
    return out


def condition_bn(sd: dict) -> dict:
    """In place: the conditioning of tools/make_conditioned.py applied by NAME to every BatchNorm of a state_dict (a prefix that owns a
    `running_mean`): weight ~ U(0.4, 0.6), bias ~ N(1.0, 0.3), each from a generator seeded by crc32(prefix).  Pre-activations get a
    positive mean, SiLU works in its near-linear range and fp32 evaluation-order noise is no longer amplified layer after layer (the
    plain name-seeded L-scale config-5 network differs from its own fp64 evaluation by 12 % after the first A2C2f).  The matching
    running statistics are calibrated by the fixture generator and stored in the fixture."""
    import zlib

    for k in [k for k in sd if k.endswith(".running_mean")]:
        p = k[: -len(".running_mean")]
        g = torch.Generator().manual_seed((zlib.crc32(p.encode()) ^ 0x5EED) & 0x7FFFFFFF)
        sd[p + ".weight"] = torch.rand(sd[p + ".weight"].shape, generator=g) * 0.2 + 0.4
        sd[p + ".bias"] = torch.randn(sd[p + ".bias"].shape, generator=g) * 0.3 + 1.0
    return sd


def cfg5_imbalance(sd: dict, alpha_image: float, alpha_token: float) -> dict:
    """The expert-imbalance knob of BASELINE config 5: yolo_master_amd.weights.expert_imbalance (bench.py --imbalance
    applies the same function)."""
    from yolo_master_amd.weights import expert_imbalance

    return expert_imbalance(sd, alpha_image, alpha_token)


def dense_pred(B: int, nc: int, A: int, seed: int, frame: float = 640.0) -> torch.Tensor:
    """A dense-scene prediction tensor y [B, 4 + nc, A] for the validator's NMS settings (conf 0.001, multi_label): EVERY (anchor,
    class) pair is a candidate (A * nc per image: 672 000 at 8400 x 80) and all scores of an image are distinct float32 values (a
    random permutation of an arithmetic sequence in (0.001, 0.999)), because the reference's unstable argsort leaves the order of
    equal scores undefined.  Boxes cluster around a few hundred objects so that greedy suppression has work.  Regenerated from the
    seed on both sides (numpy Generator: the stream is platform independent); the fixture stores only the reference's result."""
    rng = np.random.default_rng(seed)
    y = np.empty((B, 4 + nc, A), np.float32)
    for b in range(B):
        k = max(A // 24, 1)
        centres = rng.uniform(40, frame - 40, (k, 2))
        sizes = rng.uniform(24, 140, (k, 2))
        pick = rng.integers(0, k, A)
        y[b, 0:2] = (centres[pick] + rng.normal(0, 5, (A, 2))).T
        y[b, 2:4] = (sizes[pick] * rng.uniform(0.85, 1.15, (A, 2))).T
        n = nc * A
        s = (rng.permutation(n).astype(np.float64) + 1.0) / (n + 2.0) * 0.997 + 0.0015
        y[b, 4:] = s.astype(np.float32).reshape(nc, A)
        assert len(np.unique(y[b, 4:])) == n
    return torch.from_numpy(y)


def decode_best_then_nms(dev, levels=((5, 7, 8.0), (3, 4, 16.0)), B=2, nc=11, ties=False, seed=0):
    """The decode kernel's per-anchor best class (ymk_detect_decode: best_conf / best_cls) and the NMS that takes it instead of reading
    the class rows (ymk_nms_batched).  Checks, all exact: (1) y is the same with and without the side outputs, (2) they are amax /
    first argmax of y's class rows, (3) the detections and kept anchors through `y.best` equal those of the plain y and the oracle's,
    with and without a class filter, (4) an in-place edit of y drops the side outputs (the version stamp), (5) multi_label ignores
    them.  ties: class logits quantised so that several classes share the maximum (the first one must win)."""
    from oracle import nms_ref
    from yolo_master_amd import ops
    from yolo_master_amd.nms import non_max_suppression

    A = sum(h * w for h, w, _ in levels)
    y0 = torch.zeros((B, 4 + nc, A), dtype=torch.float32, device=dev)
    y1 = torch.zeros_like(y0)
    best = (torch.full((B, A), -1.0, device=dev), torch.full((B, A), -1, dtype=torch.int32, device=dev))
    off = 0
    for li, (h, w, stride) in enumerate(levels):
        box = rnd(B, h, w, 64, seed=seed + 10 * li + 1, scale=2.0).to(dev)
        cls = rnd(B, h, w, nc, seed=seed + 10 * li + 2, scale=2.0)
        if ties:
            cls = (cls * 2).round() / 2
        cls = cls.to(dev)
        ops.detect_decode(box, cls, y0, stride, off, 16)
        ops.detect_decode(box, cls, y1, stride, off, 16, best=best)
        off += h * w
    assert torch.equal(y0, y1), "y changes with the side outputs requested"
    conf, j = y1[:, 4:].cpu().max(1)
    assert torch.equal(best[0].cpu(), conf), "best_conf is not the class maximum stored in y"
    assert torch.equal(best[1].cpu().long(), j), "best_cls is not the first arg max"
    if ties:
        assert int((y1[:, 4:].cpu() == conf.unsqueeze(1)).sum(1).max()) > 1, "no shared maxima in this draw"
    y1.best = (best[0], best[1], y1._version)
    calls = []
    real = ops.lib.ymk_nms_batched

    def spy(*a):
        calls.append(a[13] is not None and a[14] is not None)   # best_conf, best_cls arguments
        return real(*a)

    class _Lib:
        def __getattr__(self, k):
            return spy if k == "ymk_nms_batched" else getattr(lib0, k)

    lib0, ops.lib = ops.lib, _Lib()
    try:
        for kw in (dict(conf_thres=0.3, iou_thres=0.5, max_det=40), dict(conf_thres=0.3, iou_thres=0.5, classes=[0, 3, 4, 9], max_det=40),
                   dict(conf_thres=0.05, iou_thres=0.6, max_nms=50, max_det=20)):
            ref, ref_idx = nms_ref.non_max_suppression(y0.cpu().numpy(), return_idxs=True, **kw)
            a, ai = non_max_suppression(y0, return_idxs=True, **kw)
            b, bi = non_max_suppression(y1, return_idxs=True, **kw)
            assert calls[-2:] == [False, True], calls
            for i in range(B):
                assert np.array_equal(bi[i].cpu().numpy(), ref_idx[i]) and np.array_equal(b[i].cpu().numpy(), ref[i]), (kw, i)
                assert torch.equal(a[i], b[i]) and torch.equal(ai[i], bi[i])
            assert sum(len(r) for r in ref) > 0
        non_max_suppression(y1, 0.3, 0.5, multi_label=True)
        assert calls[-1] is False, "multi_label must read the class rows"
        y1[:, 4:, : A // 2] *= 0.5                                   # the caller edits y: the side outputs no longer describe it
        ref, ref_idx = nms_ref.non_max_suppression(y1.cpu().numpy(), 0.3, 0.5, return_idxs=True)
        b, bi = non_max_suppression(y1, 0.3, 0.5, return_idxs=True)
        assert calls[-1] is False, "stale side outputs were used"
        for i in range(B):
            assert np.array_equal(bi[i].cpu().numpy(), ref_idx[i]) and np.array_equal(b[i].cpu().numpy(), ref[i])
    finally:
        ops.lib = lib0
