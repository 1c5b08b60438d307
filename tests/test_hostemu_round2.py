"""The three kernels rewritten in the second half of round 2, on the CPU lane emulator (tests/hostemu: unmodified kernel sources
compiled for the host, lanes as fibers) through the product's own wrappers: the table-driven ES-MoE pointwise stage
(`moe_pw_lean_kernel`, csrc/esmoe.hip) and the streaming kernel it falls back to, the branch-free resident area attention
(csrc/attn.hip: compile-time chunks, register-swap reductions, packed V^T staging) and the radix ordering of NMS candidates
(csrc/nms.hip).  Logic only (indexing, tables, reductions, barriers): speed and occupancy are the GPU tests' business."""
import numpy as np
import pytest
import torch

from tests import emu_ops


@pytest.fixture
def host_ops(hostlib, monkeypatch):
    from yolo_master_amd import ops

    monkeypatch.setattr(ops, "lib", hostlib)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    return ops


def _rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


PW_CASES = [
    # dtype, B, H, W, C, Cout, sel
    (torch.bfloat16, 3, 14, 18, 128, 128, [[0, 2], [1, -1], [-1, -1]]),     # two K groups; ragged last tile; one / two / no expert
    (torch.bfloat16, 2, 16, 16, 256, 256, [[3, 1], [2, -1]]),                # four K groups, two cout tiles, one weight slot
    (torch.bfloat16, 5, 8, 16, 64, 128, [[0, 1], [1, 0], [2, 3], [3, -1], [0, 2]]),   # one K group, four resident weight tiles
    (torch.float32, 2, 10, 13, 64, 128, [[1, 3], [0, -1]]),                  # fp32: two K groups of 32
    (torch.bfloat16, 2, 12, 12, 192, 192, [[0, 1], [2, -1]]),                # cout not a multiple of 128: the older streaming kernel
]


@pytest.mark.parametrize("case", PW_CASES, ids=lambda c: f"{str(c[0])[6:]}-{c[4]}to{c[5]}-{c[2]}x{c[3]}")
def test_esmoe_pointwise_stage(case, host_ops):
    dtype, B, H, W, C, Cout, sel = case
    E, top_k = 4, 2
    sel = torch.tensor(sel, dtype=torch.int32)
    dw_out = _rnd(B * top_k, H, W, C, seed=1).to(dtype)
    vec = 8 if dtype == torch.bfloat16 else 4
    kp = (C + 63) // 64 * 64
    pw_w = torch.zeros(E, Cout, kp, dtype=dtype)
    pw_w[:, :, :C] = _rnd(E, Cout, C, seed=2, scale=C ** -0.5).to(dtype)
    pw_b, nscale, nshift = _rnd(E, Cout, seed=3, scale=0.3), 1.0 + _rnd(Cout, seed=4, scale=0.1), _rnd(Cout, seed=5, scale=0.2)
    gate = torch.zeros(B, E)
    for b in range(B):
        for k in range(top_k):
            if sel[b, k] >= 0:
                gate[b, sel[b, k]] = 0.3 + 0.5 * torch.rand(1, generator=torch.Generator().manual_seed(10 * b + k)).item()
    ref = emu_ops.esmoe_pw(dw_out, B, H, W, pw_w, pw_b, nscale, nshift, top_k, sel, gate)
    out = torch.full((B, H, W, Cout + 8), 7.0, dtype=dtype)[..., :Cout]   # a view with a row pitch: ldy != Cout
    got = host_ops.esmoe_pw(dw_out, B, H, W, pw_w, pw_b, nscale, nshift, top_k, sel, gate, out=out)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert torch.allclose(got.float(), ref.float(), atol=tol, rtol=tol), float((got.float() - ref.float()).abs().max())
    assert bool((out.storage_offset() == 0)) and torch.all(out.as_strided((B, H, W, 8), out.stride(), Cout) == 7.0), "wrote past the channel range"


ATTN_CASES = [
    # dtype, B, N, heads, area
    (torch.bfloat16, 1, 400, 2, 1),    # the detector's 400 keys: 13 key-tile pairs -> chunks of 7 + 6, -inf mask in the last pair
    (torch.bfloat16, 2, 144, 1, 2),    # 72 keys per area: one chunk of three pairs, ragged
    (torch.bfloat16, 1, 32, 1, 1),     # a single pair, no mask
    (torch.float32, 1, 48, 2, 1),      # fp32 path of the same code (per-tile P V)
    (torch.bfloat16, 1, 1100, 1, 1),   # more keys than the resident kernel holds: the 256-query kernel of csrc/mixattn.hip (18 key blocks, ragged)
    (torch.bfloat16, 2, 2200, 2, 2),   # ... with areas as batch entries
]


@pytest.mark.parametrize("case", ATTN_CASES, ids=lambda c: f"{str(c[0])[6:]}-N{c[2]}-h{c[3]}-a{c[4]}")
def test_area_attention(case, host_ops):
    dtype, B, N, heads, area = case
    qkv = _rnd(B, 1, N, 3 * heads * 32, seed=7, scale=0.8).to(dtype)
    ref = emu_ops.area_attn(qkv, heads, area)
    got = host_ops.area_attn(qkv, heads, area)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert torch.allclose(got.float(), ref.float(), atol=tol, rtol=tol), float((got.float() - ref.float()).abs().max())


QKV_ATTN_CASES = [
    # B, H, W, heads, area, x pad, out pad
    (1, 20, 20, 4, 1, 0, 0),      # the detector's 400-token area at C = 128: 25 tiles over four waves (7 / 6 / 6 / 6), two chunks of key-tile pairs
    (2, 6, 12, 2, 2, 64, 64),     # C = 64 (two k-steps), 36 tokens per area: ragged last token tile AND ragged key-tile pair, strided views
    (1, 4, 8, 4, 1, 0, 128),      # a single key-tile pair, fewer token tiles than waves
]


def run_qkv_attn_case(ops, case, dev="cpu", dtype=torch.bfloat16):
    """ymk_area_attn_qkv against the composition it replaces (1x1 convolution storing 16-bit qkv, then ymk_area_attn's arithmetic restated
    in fp32) on the same 16-bit operands.  Shared with tests/test_gpu_kernels.py."""
    from tests import emu_ops

    B, H, W, heads, area, xpad, opad = case
    C = heads * 32
    x = _rnd(B, H, W, C, seed=5 + H, scale=1.0).to(dtype)
    w = _rnd(3 * C, C, seed=6, scale=1.5 * C ** -0.5).to(dtype)          # rows already in [Q | K | V] order
    b = _rnd(3 * C, seed=7, scale=0.3)
    ref_o, ref_v = emu_ops.area_attn_qkv(x, w, b, heads, area)
    xb = torch.full((B, H, W, C + xpad), 3.0, dtype=dtype)
    xb[..., xpad // 2: xpad // 2 + C] = x
    xd = xb.to(dev)[..., xpad // 2: xpad // 2 + C]
    ob = torch.full((B, H, W, C + opad), 7.0, dtype=dtype, device=dev)
    vb = torch.full((B, H, W, C + opad), 9.0, dtype=dtype, device=dev)
    o, v = ob[..., opad // 2: opad // 2 + C], vb[..., opad // 2: opad // 2 + C]
    assert ops.area_attn_qkv_supported(dtype, C, heads, H * W, area)
    got_o, got_v = ops.area_attn_qkv(xd, w.to(dev), b.to(dev), heads, area, out=o, v_out=v)
    assert got_o.data_ptr() == o.data_ptr() and got_v.data_ptr() == v.data_ptr()
    # v: one rounding of the same fp32 sums (summation order may differ in the last bit of the 16-bit value)
    ev = (got_v.float().cpu() - ref_v.float()).abs()
    assert float(ev.max()) <= 2.0 ** -6 * max(1.0, float(ref_v.float().abs().max())), f"{case}: v max err {float(ev.max()):.3e}"
    assert float((ev > 0).float().mean()) <= 0.02, f"{case}: {float((ev > 0).float().mean()):.4f} of v differ"
    eo = (got_o.float().cpu() - ref_o.float()).abs()
    assert float(eo.max()) <= 3e-2 * max(1.0, float(ref_o.float().abs().max())), f"{case}: attention max err {float(eo.max()):.3e}"
    assert float(eo.mean()) <= 2e-3, f"{case}: attention mean err {float(eo.mean()):.3e}"
    if opad:
        for buf, fill in ((ob, 7.0), (vb, 9.0)):
            assert bool((buf[..., : opad // 2].float().cpu() == fill).all()) and bool((buf[..., opad // 2 + C:].float().cpu() == fill).all()), "wrote outside the view"


@pytest.mark.parametrize("case", QKV_ATTN_CASES, ids=lambda c: f"{c[1]}x{c[2]}-h{c[3]}-a{c[4]}")
def test_area_attention_with_the_projection_inside(case, host_ops):
    run_qkv_attn_case(host_ops, case)


@pytest.mark.parametrize("ties", [False, True])
def test_nms_radix_ordering(ties, host_ops):
    """More than 1024 candidates per image -> the radix index sort; kept anchors and detections bit-exact against the oracle,
    with heavy score ties (17 distinct scores) and without."""
    from oracle import nms_ref
    from yolo_master_amd.nms import non_max_suppression

    g = torch.Generator().manual_seed(31)
    B, nc, A = 2, 3, 1700
    xy = torch.rand(B, 2, A, generator=g) * 600 + 20
    wh = torch.rand(B, 2, A, generator=g) * 50 + 4
    cls = torch.rand(B, nc, A, generator=g) * 0.6 + 0.2
    cls[1, :, 1100:] *= 0.1                      # the second image stays under 1024 candidates... or not: both paths in one launch
    if ties:
        cls = (cls * 16).round() / 16
    y = torch.cat([xy, wh, cls], 1)
    ref, ref_idx = nms_ref.non_max_suppression(y.numpy(), 0.25, 0.6, return_idxs=True, max_det=60)
    got, got_idx = non_max_suppression(y, 0.25, 0.6, return_idxs=True, max_det=60)
    for b in range(B):
        assert np.array_equal(got_idx[b].numpy(), ref_idx[b]), f"image {b}: kept anchors differ"
        assert np.array_equal(got[b].numpy(), ref[b]), f"image {b}: detections differ"


@pytest.mark.parametrize("mode", ["multi", "multi_ties", "single", "classes", "mixed_batch"])
def test_nms_selects_the_top_max_nms_candidates(mode, host_ops):
    """More candidates than `max_nms` (utils/nms.py:142-146 keeps the max_nms best by score, whatever their number): the radix select
    on the score bits (csrc/nms.hip, three histogram levels) + threshold-aware compaction, bit-exact against the oracle's
    sort-and-truncate.  `multi_ties`: 33 distinct scores, so the threshold value itself is shared by hundreds of candidates and only
    the first few (anchor-major order, what the stable sort keeps) may pass; `mixed_batch`: one image over the cap, one under it."""
    from oracle import nms_ref
    from yolo_master_amd.nms import non_max_suppression

    g = torch.Generator().manual_seed(77)
    B, nc, A = 2, 6, 900
    xy = torch.rand(B, 2, A, generator=g) * 600 + 20
    wh = torch.rand(B, 2, A, generator=g) * 60 + 8
    cls = torch.rand(B, nc, A, generator=g) * 0.9 + 0.05
    kw = dict(conf_thres=0.01, iou_thres=0.6, multi_label=mode != "single", max_det=80, max_nms=700)
    if mode == "multi_ties":
        cls = (cls * 32).round() / 32
    if mode == "single":
        kw["max_nms"] = 300                      # 900 best-class candidates > 300
    if mode == "classes":
        kw["classes"] = [0, 2, 5]
    if mode == "mixed_batch":
        cls[1] *= 0.05                           # image 1: ~500 candidates above conf, under the cap
    y = torch.cat([xy, wh, cls], 1)
    ref, ref_idx = nms_ref.non_max_suppression(y.numpy(), return_idxs=True, **kw)
    got, got_idx = non_max_suppression(y, return_idxs=True, **kw)
    ncand = [(int((cls[b] > 0.01).sum()) if mode != "single" else int((cls[b].amax(0) > 0.01).sum())) for b in range(B)]
    assert ncand[0] > kw["max_nms"], ncand
    for b in range(B):
        assert np.array_equal(got_idx[b].numpy(), ref_idx[b]), f"{mode} image {b}: kept anchors differ ({ncand[b]} candidates)"
        assert np.array_equal(got[b].numpy(), ref[b]), f"{mode} image {b}: detections differ"


@pytest.mark.parametrize("case", ["seg", "seg_multi", "seg_caps"])
def test_nms_with_carried_mask_rows_vs_reference_golden(case, host_ops, golden_dir):
    """The segment predictor's NMS call (models/yolo/segment/predict.py -> detect/predict.py:54-65: `nc=len(names)`, so the rows
    behind the class rows are mask coefficients that ride along, utils/nms.py:76-81,117,122,127): csrc/nms.hip with `extra` rows +
    ymk_nms_gather_rows on the lane emulator, bit-exact against the REAL reference's output rows (xyxy, conf, cls, mask...)."""
    from tests.helpers import load_npz
    from yolo_master_amd.nms import non_max_suppression

    z = load_npz(golden_dir / f"nms_{case}.npz")
    kw = dict(conf_thres=float(z["arg_conf_thres"]), iou_thres=float(z["arg_iou_thres"]), multi_label=bool(z["arg_multi_label"]),
              agnostic=bool(z["arg_agnostic"]), max_det=int(z["arg_max_det"]), max_nms=int(z["arg_max_nms"]), nc=int(z["arg_nc"]))
    if "arg_classes" in z:
        kw["classes"] = z["arg_classes"].tolist()
    y = torch.from_numpy(z["y"])
    got, idx = non_max_suppression(y, return_idxs=True, **kw)
    extra = y.shape[1] - 4 - kw["nc"]
    for b in range(y.shape[0]):
        assert got[b].shape[1] == 6 + extra
        assert np.array_equal(idx[b].numpy(), z[f"idx{b}"]), f"{case} image {b}: kept anchors differ"
        assert np.array_equal(got[b].numpy(), z[f"dets{b}"]), f"{case} image {b}: output rows differ"


@pytest.mark.parametrize("ties", [False, True])
def test_decode_side_outputs_feed_nms(ties, host_ops):
    """Round 4: the decode kernel hands NMS every anchor's best class (tests/helpers.decode_best_then_nms), on the lane emulator."""
    from tests.helpers import decode_best_then_nms

    decode_best_then_nms("cpu", ties=ties, seed=5)


def _dw_stage_inputs(dtype, B, H, W, C, ksizes, sel, seed=3):
    """Packed depthwise filters [sum k*k, C], the CSR of retained (image, slot) pairs per expert and an input with a row pitch."""
    E, top_k = len(ksizes), len(sel[0])
    sel = torch.tensor(sel, dtype=torch.int32)
    xbuf = _rnd(B, H, W, C + 8, seed=seed).to(dtype)
    x = xbuf[..., :C]                                           # ldx != C
    offs, parts = [], []
    for e, k in enumerate(ksizes):
        offs.append(sum(p.numel() for p in parts))
        parts.append(_rnd(k * k, C, seed=seed + 10 + e, scale=1.0 / k).to(dtype).reshape(-1))
    dw_w = torch.cat(parts)
    csr_off, csr_pair = [0], []
    for e in range(E):
        csr_pair += [int(p) for p in torch.nonzero(sel.reshape(-1) == e).reshape(-1)]
        csr_off.append(len(csr_pair))
    pad = B * top_k - len(csr_pair)
    return (x, dw_w, torch.tensor(offs, dtype=torch.int32), torch.tensor(ksizes, dtype=torch.int32), sel,
            torch.tensor(csr_off, dtype=torch.int32), torch.tensor(csr_pair + [0] * pad, dtype=torch.int32))


DW_STAGE_CASES = [
    # dtype, B, H, W, C, ksizes, sel
    (torch.bfloat16, 4, 13, 23, 64, [3, 5, 7, 9], [[0, 3], [2, 1], [1, -1], [-1, -1]]),    # every (stencil, halo) combination of a pair; ragged tiles
    (torch.bfloat16, 3, 9, 44, 24, [3, 5, 7, 9], [[3, 0], [-1, 2], [1, 1]]),               # wide tiles; C not a multiple of 16; a leading dropped slot
    (torch.bfloat16, 2, 8, 21, 32, [3, 3, 5, 5], [[0, 1, 2], [3, -1, 0]]),                 # three slots per image, equal stencil sizes
    (torch.float32, 2, 10, 20, 16, [3, 5, 7, 9], [[0, 3], [2, -1]]),                       # fp32: the per-pair kernel
    (torch.bfloat16, 2, 10, 20, 16, [3, 7, 11, 5], [[2, 0], [1, 3]]),                      # an 11-tap expert: the per-pair kernel
]


@pytest.mark.parametrize("case", DW_STAGE_CASES, ids=lambda c: f"{str(c[0])[6:]}-{c[2]}x{c[3]}x{c[4]}-k{'_'.join(map(str, c[5]))}")
def test_esmoe_depthwise_stage(case, host_ops):
    """The depthwise stage of the retained (image, expert) pairs: since round 5 one workgroup stages an image tile's halo once (for the
    largest stencil among the image's retained experts) and runs every retained expert on it.  Planes of dropped slots stay untouched."""
    dtype, B, H, W, C, ksizes, sel = case
    x, dw_w, dw_off, ks, sel_t, csr_off, csr_pair = _dw_stage_inputs(dtype, B, H, W, C, ksizes, sel)
    top_k = sel_t.shape[1]
    ref = emu_ops.esmoe_dw(x, dw_w, dw_off, ks, max(ksizes), top_k, sel_t, csr_off, csr_pair)
    got = host_ops.esmoe_dw(x, dw_w, dw_off, ks, max(ksizes), top_k, sel_t, csr_off, csr_pair)
    live = (sel_t.reshape(-1) >= 0)
    tol = 2e-2 if dtype != torch.float32 else 2e-5
    err = float((got[live].float() - ref[live].float()).abs().max())
    assert err <= tol * max(1.0, float(ref[live].float().abs().max())), err


def run_nms_threshold_ties(nms_fn):
    """IoU thresholds that sit EXACTLY on, one ulp below and one ulp above the float32 IoU of a pair of the scene — and, in the same
    image, hundreds of pairs at random overlaps — must give the oracle's kept set (the reference's `inter / union > thr`, utils/nms.py).  Shared
    with tests/test_gpu_kernels.py."""
    from oracle import nms_ref

    g = torch.Generator().manual_seed(77)
    npair = 260
    cx = (torch.arange(npair) % 20).float() * 30 + 20 + torch.rand(npair, generator=g)
    cy = (torch.arange(npair) // 20).float() * 40 + 20 + torch.rand(npair, generator=g)
    w0, h0 = torch.rand(npair, generator=g) * 8 + 12, torch.rand(npair, generator=g) * 8 + 12
    dx, dy = (torch.rand(npair, generator=g) - 0.5) * 14, (torch.rand(npair, generator=g) - 0.5) * 14      # second box of the pair: shifted, own size
    w1, h1 = torch.rand(npair, generator=g) * 8 + 12, torch.rand(npair, generator=g) * 8 + 12
    xywh = torch.stack([torch.cat([cx, cx + dx]), torch.cat([cy, cy + dy]), torch.cat([w0, w1]), torch.cat([h0, h1])])       # [4, 2 npair]
    score = torch.cat([torch.rand(npair, generator=g) * 0.3 + 0.6, torch.rand(npair, generator=g) * 0.3 + 0.3])           # the first box of a pair ranks higher
    y = torch.cat([xywh, score[None]], 0)[None].contiguous()                                                                  # [1, 5, A], one class
    # float32 IoU of pair 0, computed the reference's way
    b = nms_ref.xywh2xyxy(y[0, :4].t().numpy()) if hasattr(nms_ref, "xywh2xyxy") else None
    if b is None:
        c = y[0, :4].t().numpy().astype(np.float32)
        b = np.stack([c[:, 0] - c[:, 2] / np.float32(2), c[:, 1] - c[:, 3] / np.float32(2), c[:, 0] + c[:, 2] / np.float32(2), c[:, 1] + c[:, 3] / np.float32(2)], 1)
    ious = []
    for p in (0, 1, 2, 3):
        a0, a1 = b[p], b[npair + p]
        iw = max(np.float32(0), min(a0[2], a1[2]) - max(a0[0], a1[0])); ih = max(np.float32(0), min(a0[3], a1[3]) - max(a0[1], a1[1]))
        inter = np.float32(iw * ih)
        ar0, ar1 = np.float32((a0[2] - a0[0]) * (a0[3] - a0[1])), np.float32((a1[2] - a1[0]) * (a1[3] - a1[1]))
        ious.append(np.float32(inter / np.float32(np.float32(ar0 + ar1) - inter)))
    thrs = []
    for v in ious:
        if 0.05 < v < 0.95:
            thrs += [float(v), float(np.nextafter(v, np.float32(0))), float(np.nextafter(v, np.float32(1)))]
    assert len(thrs) >= 3, ious
    for thr in thrs + [0.45, 0.7]:
        ref, ref_idx = nms_ref.non_max_suppression(y.numpy(), 0.25, thr, return_idxs=True, max_det=600)
        got, got_idx = nms_fn(y, 0.25, thr, return_idxs=True, max_det=600)
        assert np.array_equal(np.asarray(got_idx[0].cpu()), ref_idx[0]), f"iou threshold {thr!r}: kept anchors differ"
        assert np.array_equal(np.asarray(got[0].cpu()), ref[0]), f"iou threshold {thr!r}: detections differ"


def test_nms_iou_threshold_ties(host_ops):
    from yolo_master_amd.nms import non_max_suppression

    run_nms_threshold_ties(non_max_suppression)
