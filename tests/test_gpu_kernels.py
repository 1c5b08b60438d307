"""GPU parity of every libymk kernel (called through the C-ABI) against plain PyTorch fp32 CPU
references of the same op / the oracle restatement.  fp32 and bf16 compute types."""
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_close, bf16_round, module_sd, nchw, nhwc, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]   # fp16 = libymk_f16.so (the reference's half=True precision)


def _prep(t, dtype):
    return t.to(dtype).float() if dtype != torch.float32 else t


# ------------------------------------------------------------------------------- conv (MFMA igemm)
CONV_CASES = [
    # B, H, W, Cin, Cout, k, s, act, residual, pad_in, pad_out
    (2, 20, 20, 64, 128, 1, 1, True, False, 0, 0),
    (2, 20, 20, 64, 256, 3, 1, True, True, 0, 0),
    (1, 33, 29, 16, 8, 3, 1, True, False, 0, 0),      # odd sizes, tiny Cout (N-scale bottleneck)
    (2, 40, 40, 32, 64, 3, 2, True, False, 16, 32),   # stride 2, channel-slice views in and out
    (3, 17, 23, 8, 16, 3, 2, False, False, 8, 0),     # Cin = 8 (tap crosses a 16-byte chunk boundary)
    (2, 16, 16, 192, 80, 1, 1, False, False, 0, 0),   # Cout = 80 (Detect cls tail), K not multiple of 64
    (1, 12, 12, 512, 512, 3, 1, True, True, 0, 0),    # K = 4608
    (2, 9, 9, 48, 32, 1, 1, True, True, 16, 16),
    # shapes that route to the specialised kernels (see csrc/conv.hip dispatch):
    (4, 128, 128, 64, 128, 1, 1, True, True, 0, 0),    # weight-stationary streaming 1x1 (512 pixel tiles, 1 K group)
    (8, 128, 128, 192, 160, 1, 1, True, False, 64, 32),  # streaming 1x1, 3 K groups, 2 cout tiles (ragged), slice views
    (4, 96, 96, 32, 32, 3, 1, True, True, 0, 0),       # spatial-tile 3x3, Cin 32 -> 32 (LDS im2col), residual
    (3, 100, 90, 32, 64, 3, 1, True, False, 32, 0),    # spatial-tile 3x3, ragged tiles, Cout 64, input slice view
    (4, 96, 96, 64, 64, 3, 1, False, True, 0, 64),     # spatial-tile 3x3, Cin 64 (bf16 only; fp32 falls back)
    (4, 96, 96, 64, 16, 3, 1, True, False, 0, 0),      # spatial-tile 3x3, Cout 16 inside a 32-wide tile
    (4, 96, 90, 16, 32, 3, 1, True, True, 16, 0),      # spatial-tile 3x3, Cin 16 (bf16: two taps per MFMA), residual, slice view
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, dtype):
    from yolo_master_amd import ops

    B, H, W, Cin, Cout, k, s, act, res, pin, pout = case
    x = _prep(rnd(B, Cin, H, W, seed=1), dtype)
    w = _prep(rnd(Cout, Cin, k, k, seed=2, scale=(1.0 / (Cin * k * k)) ** 0.5), dtype)
    b = rnd(Cout, seed=3, scale=0.1)
    ref = F.conv2d(x, w, b, s, k // 2)
    if act:
        ref = F.silu(ref)
    Ho, Wo = ref.shape[2:]
    r = _prep(rnd(B, Cout, Ho, Wo, seed=4), dtype) if res else None
    if res:
        ref = r + ref
    xd = nhwc(x, dtype, DEV, pin, pin // 2)
    out_buf = torch.zeros((B, Ho, Wo, Cout + pout), dtype=dtype, device=DEV)
    out = out_buf[..., pout // 2: pout // 2 + Cout]
    wp = ops.pack_conv_weight(w.to(DEV), dtype)
    y = ops.conv2d(xd, wp, b.to(DEV), k, s, act, out=out, residual=nhwc(r, dtype, DEV) if res else None)
    torch.cuda.synchronize()
    assert_close(nchw(y), ref, dtype, f"conv2d {case}")
    if pout:  # the neighbouring channels of the wider buffer must be untouched
        assert float(out_buf[..., : pout // 2].abs().max()) == 0.0
        assert float(out_buf[..., pout // 2 + Cout:].abs().max()) == 0.0


def test_conv2d_f32_out_from_bf16():
    from yolo_master_amd import ops

    x = bf16_round(rnd(2, 64, 10, 10, seed=1))
    w = bf16_round(rnd(64, 64, 1, 1, seed=2, scale=0.125))
    b = rnd(64, seed=3)
    y = ops.conv2d(nhwc(x, torch.bfloat16, DEV), ops.pack_conv_weight(w.to(DEV), torch.bfloat16), b.to(DEV), 1, 1, False,
                   out_dtype=torch.float32)
    assert y.dtype == torch.float32
    assert_close(nchw(y), F.conv2d(x, w, b), torch.float32, "bf16 conv with fp32 output")


@pytest.mark.parametrize("dtype", DTYPES)
def test_mfma_layout_asymmetric(dtype):
    """A = I against an asymmetric B: catches row/column swaps of the MFMA result layout."""
    from yolo_master_amd import ops

    C = 64
    x = torch.arange(2 * C * 5 * 7, dtype=torch.float32).reshape(2, C, 5, 7) % 13 - 6.0
    w = torch.zeros(C, C, 1, 1)
    for o in range(C):
        w[o, (o * 7 + 3) % C, 0, 0] = 1.0 + (o % 3)  # permutation * asymmetric scale
    ref = F.conv2d(x, w)
    y = ops.conv2d(nhwc(x, dtype, DEV), ops.pack_conv_weight(w.to(DEV), dtype), torch.zeros(C, device=DEV), 1, 1, False)
    assert torch.equal(nchw(y), ref), "MFMA fragment/result layout is wrong (exact integer test)"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 64, 48, 16, True),    # W % 4 == 0: LDS-staged rows kernel
                                   (3, 70, 52, 32, True),    # ragged row blocks (Ho = 35), Cout 32
                                   (2, 33, 30, 16, True),    # W % 4 != 0: direct-gather MFMA kernel
                                   (2, 64, 48, 16, False)])  # no K-major weights: generic VALU kernel
def test_stem(dtype, shape):
    from yolo_master_amd import ops

    B, H, W, co, kmajor = shape
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(5))
    w = rnd(co, 3, 3, 3, seed=6, scale=0.3)
    b = rnd(co, seed=7, scale=0.1)
    ref = F.silu(F.conv2d(x, w, b, 2, 1))
    wp = w.permute(0, 2, 3, 1).reshape(co, -1).contiguous().to(DEV)   # [Cout][(ky,kx,c)]
    wt = wp.t().contiguous() if kmajor else None                       # [(ky,kx,c)][Cout]
    y = ops.conv2d_stem(x.to(DEV), wp, b.to(DEV), 3, 2, True, dtype, wt=wt)
    assert_close(nchw(y), ref, dtype, "stem conv")


# ------------------------------------------------------------------------------- depthwise
DW_STAGE = [
    # B, H, W, C, ksizes, sel
    (4, 40, 60, 64, [3, 5, 7, 9], [[0, 3], [2, 1], [1, -1], [-1, -1]]),     # every (stencil, halo) combination of a pair; 20-wide tiles
    (3, 80, 80, 128, [3, 5, 7, 9], [[3, 0], [-1, 2], [1, 1]]),              # a leading dropped slot, one expert twice
    (2, 160, 160, 24, [3, 5, 7, 9], [[2, 3], [0, 1]]),                      # 40-wide tiles; C not a multiple of 16
    (2, 33, 21, 32, [3, 3, 5, 5], [[0, 1, 2], [3, -1, 0]]),                 # three slots per image
    (2, 20, 20, 16, [3, 7, 11, 5], [[2, 0], [1, 3]]),                       # an 11-tap expert: the per-pair kernel
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("case", DW_STAGE, ids=lambda c: f"{c[1]}x{c[2]}x{c[3]}-k{'_'.join(map(str, c[4]))}")
def test_esmoe_depthwise_stage_vs_per_expert_dwconv(case, dtype):
    """ymk_esmoe_dw (one workgroup per retained (image, expert) pair and tile, the stencil size read from the selection) against
    ymk_dwconv run once per retained pair: the same stencil code on the same data.  fp32 identical; 16-bit within 1 ulp on <= 0.1 % of
    the outputs (the bound a variant that pairs its taps for v_dot2 at the other parity would need; today's kernels agree exactly)."""
    from tests.test_hostemu_round2 import _dw_stage_inputs
    from yolo_master_amd import ops

    B, H, W, C, ksizes, sel = case
    x, dw_w, dw_off, ks, sel_t, csr_off, csr_pair = (t.to(DEV) for t in _dw_stage_inputs(dtype, B, H, W, C, ksizes, sel))
    top_k = sel_t.shape[1]
    got = ops.esmoe_dw(x, dw_w, dw_off, ks, max(ksizes), top_k, sel_t, csr_off, csr_pair)
    zero_b = torch.zeros(C, device=DEV)
    for b in range(B):
        for j in range(top_k):
            e = sel[b][j]
            if e < 0:
                continue
            k = ksizes[e]
            w = dw_w[int(dw_off[e]): int(dw_off[e]) + k * k * C].reshape(k * k, C).contiguous()
            ref = ops.dwconv2d(x[b: b + 1], w, zero_b, k, False)
            g, r = got[b * top_k + j].float(), ref[0].float()
            if dtype == torch.float32:
                assert torch.equal(g, r), f"image {b} slot {j} expert {e} (k={k})"
            else:
                ulp = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * r.abs().clamp_min(2.0 ** -10)
                assert bool(((g - r).abs() <= ulp).all()), f"image {b} slot {j} expert {e} (k={k}): {float((g - r).abs().max())}"
                assert float((g != r).float().mean()) <= 1e-3, f"image {b} slot {j} expert {e} (k={k})"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [3, 5, 7, 9, 15])
def test_dwconv(k, dtype):
    from yolo_master_amd import ops

    B, C, H, W = 2, 24, 19, 21
    x = _prep(rnd(B, C, H, W, seed=1), dtype)
    w = _prep(rnd(C, 1, k, k, seed=2, scale=1.0 / k), dtype)
    b = rnd(C, seed=3, scale=0.1)
    r = _prep(rnd(B, C, H, W, seed=4), dtype)
    ref = r + F.silu(F.conv2d(x, w, b, 1, k // 2, 1, C))
    y = ops.dwconv2d(nhwc(x, dtype, DEV, 8, 4), ops.pack_dw_weight(w.to(DEV), dtype), b.to(DEV), k, True,
                     residual=nhwc(r, dtype, DEV))
    assert_close(nchw(y), ref, dtype, f"dwconv k={k}")




@pytest.mark.parametrize("case", __import__("tests.test_hostemu_mlp", fromlist=["CASES"]).CASES + [(128, 256, 102400, 0, 0), (256, 512, 25600, 0, 0)])
def test_mlp_fused(case):
    """Fused ABlock MLP kernel (csrc/mlp.hip) through the C-ABI against the two-GEMM composition in fp32 (incl. the detector's sizes)."""
    from tests.test_hostemu_mlp import run_case
    from yolo_master_amd import _lib

    run_case(_lib.load(), case, dev=DEV, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", __import__("tests.test_hostemu_round2", fromlist=["QKV_ATTN_CASES"]).QKV_ATTN_CASES
                         + [(16, 40, 40, 4, 4, 0, 0), (3, 20, 40, 4, 2, 128, 0), (4, 16, 28, 2, 1, 0, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_area_attn_qkv_fused(case, dtype):
    """The qkv projection inside the attention kernel (csrc/attn.hip area_attn_qkv_kernel) through the C-ABI against the 1x1 convolution +
    attention composition in fp32 on the same 16-bit operands (incl. the detector's 40x40 / area 4 / 128-channel shape and a 448-token area)."""
    from tests.test_hostemu_round2 import run_qkv_attn_case
    from yolo_master_amd import ops

    run_qkv_attn_case(ops, case, dev=DEV, dtype=dtype)
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------- layout kernels
@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_kernels(dtype):
    from yolo_master_amd import ops

    x = _prep(rnd(2, 16, 5, 6, seed=1), dtype)
    xd = nhwc(x, dtype, DEV, 8, 8)
    up = ops.upsample2x(xd)
    assert torch.equal(nchw(up), F.interpolate(x, scale_factor=2.0, mode="nearest"))
    buf = torch.zeros((2, 5, 6, 40), dtype=dtype, device=DEV)
    ops.copy_channels(xd, buf[..., 8:24])
    assert torch.equal(nchw(buf[..., 8:24]), x) and float(buf[..., :8].abs().max()) == 0.0
    assert torch.equal(ops.nhwc_to_nchw_f32(xd).cpu(), x)


# ------------------------------------------------------------------------------- module blocks vs oracle
def _run_module(mod, x, dtype):
    from yolo_master_amd.nn.modules import set_compute_dtype

    mod.eval().to(DEV)
    set_compute_dtype(mod, dtype)
    with torch.inference_mode():
        return mod(x.to(DEV)).float().cpu()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c3k", [False, True])
def test_c3k2_block(c3k, dtype):
    from oracle import model_ref
    from yolo_master_amd.nn.modules import C3k2

    m = C3k2(64, 128, 2, c3k, 0.5 if c3k else 0.25)
    sd = module_sd(m)
    x = rnd(2, 64, 20, 24, seed=9)
    with torch.inference_mode():
        ref = model_ref.c3k2(sd, "model.0", _prep(x, dtype))
    assert_close(_run_module(m, x, dtype), ref, dtype, f"C3k2 c3k={c3k}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("area,hw,residual", [(4, (20, 20), False), (1, (10, 12), False), (4, (4, 4), False),
                                               (4, (8, 12), True)])   # l/x scales: gamma-residual, mlp_ratio 1.2
def test_a2c2f_block(area, hw, residual, dtype):
    from oracle import model_ref
    from yolo_master_amd.nn.modules import A2C2f

    m = A2C2f(128, 128, 2, True, area, residual, 1.2 if residual else 2.0)
    sd = module_sd(m)
    if residual:
        sd["model.0.gamma"] = 0.5 + rnd(128, seed=21, scale=0.2)   # the 0.01 init would hide the branch
        m.load_state_dict({k[len("model.0."):]: v for k, v in sd.items()})
    x = rnd(2, 128, *hw, seed=10)
    with torch.inference_mode():
        ref = model_ref.a2c2f(sd, "model.0", _prep(x, dtype), area)
    assert_close(_run_module(m, x, dtype), ref, dtype, f"A2C2f area={area} hw={hw}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,heads,area", [(3, 1600, 4, 4),      # the detector's layer 8: 4 areas of 400 tokens (resident K/V)
                                            (2, 400, 8, 1),       # layer 11
                                            (2, 37 * 3, 2, 3),    # ragged: 37 tokens per area (one partial query / key tile)
                                            (1, 1024, 1, 1),      # largest resident bf16 case, 4 key chunks
                                            (1, 1100, 2, 1)])     # past the resident limit: streaming kernel
def test_area_attn_kernel(dtype, B, N, heads, area):
    """ymk_area_attn against softmax(q k^T / sqrt(32)) v per (image, area, head) in fp32 on the same (rounded) inputs."""
    from yolo_master_amd import ops

    Cq = heads * 32
    qkv = _prep(rnd(B, N, 1, 3 * Cq, seed=31), dtype)             # [B, tokens, 1, Q|K|V]
    dev = qkv.to(dtype).to(DEV)
    out = ops.area_attn(dev, heads, area)
    Na = N // area
    q, k, v = (qkv[..., i * Cq:(i + 1) * Cq].reshape(B, area, Na, heads, 32).permute(0, 1, 3, 2, 4) for i in range(3))
    att = torch.softmax((q @ k.transpose(-1, -2)) * 32 ** -0.5, -1) @ v                # [B, area, heads, Na, 32]
    ref = att.permute(0, 1, 3, 2, 4).reshape(B, N, 1, Cq)
    assert_close(out.float().cpu(), ref, dtype, f"area_attn N={N} heads={heads} area={area}", scale_aware=True)


@pytest.mark.parametrize("dtype", DTYPES)
def test_area_attention_long(dtype):
    """More than one 256-key chunk (online softmax across chunks): 24x24 = 576 tokens, area 1."""
    from oracle import model_ref
    from yolo_master_amd.nn.modules import ABlock

    m = ABlock(64, 2, 2.0, 1)
    sd = module_sd(m)
    x = rnd(1, 64, 24, 24, seed=11)
    with torch.inference_mode():
        ref = model_ref.ablock(sd, "model.0", _prep(x, dtype), 1)
    assert_close(_run_module(m, x, dtype), ref, dtype, "ABlock 576 tokens")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("chunked,cin,hw", [(False, 64, (14, 18)), (True, 128, (14, 18)),   # chunked: the two expert stages walked in image chunks (ES_MOE.chunk_mb)
                                            (False, 128, (14, 18)),
                                            (False, 256, (14, 18)),    # table-driven pointwise stage, 2 cout tiles, 4 K groups
                                            (True, 256, (14, 18)),
                                            (False, 192, (14, 18)),    # cout not a multiple of 128: the older streaming kernel
                                            (False, 128, (88, 88))])   # more (image, tile) items than workgroups, ragged last tile
def test_esmoe_block(dtype, chunked, cin, hw):
    from oracle import model_ref
    from yolo_master_amd.nn.modules import ES_MOE

    m = ES_MOE(cin, cin)
    if chunked:   # 6 images in chunks of 2 (the depthwise planes of two images at a time): per image the same kernels on the same data
        m.chunk_mb = 2 * 2 * hw[0] * hw[1] * cin * (4 if dtype == torch.float32 else 2) / 1e6
    sd = module_sd(m)
    # per-image offsets so that images route differently
    x = rnd(6, cin, hw[0], hw[1], seed=12) + rnd(6, cin, 1, 1, seed=13, scale=1.5)
    info = {}
    with torch.inference_mode():
        ref = model_ref.es_moe(sd, "model.0", _prep(x, dtype), info=info)
    y = _run_module(m, x, dtype)
    r = info["model.0"]
    got = m.last_route
    retained = (got["gate_w"] > 0).cpu()
    if dtype == torch.float32:
        assert torch.equal(retained, r["retained"]), "retained expert set differs from the oracle"
        assert_close(got["route_w"], r["route_w"], dtype, "routing weights")
        assert_close(got["gate_w"], r["gate_w"], dtype, "gate weights")
    if torch.equal(retained, r["retained"]):
        assert_close(y, ref, dtype, "ES_MOE output")
    # CSR permutation invariants
    off, pair, sel = got["csr_off"].cpu().tolist(), got["csr_pair"].cpu().tolist(), got["sel"].cpu()
    E, top_k = 4, 2
    assert off[0] == 0 and off[E] == int((sel >= 0).sum())
    for e in range(E):
        seg = pair[off[e]:off[e + 1]]
        assert seg == sorted(seg) and all(int(sel.view(-1)[p]) == e for p in seg)


ESMOE_MODES = {"sparse": {}, "dense": dict(use_sparse_inference=False), "disabled": {}, "all": dict(top_k=None),
               "k3of4": dict(top_k=3, dynamic_threshold=0.2)}


@pytest.mark.parametrize("case", list(ESMOE_MODES))
def test_esmoe_modes_and_state_vs_reference_golden(case, golden_dir):
    """Dispatch modes of ES_MOE (sparse / dense over the top-k set / top_k=None) and the eval-time buffers
    `expert_usage_counts` / `load_balancing_loss` (modules.py:706-741), against vectors from the REAL reference module
    (tests/golden/make_golden_esmoe.py).  fp32."""
    from tests.helpers import load_npz
    from yolo_master_amd.nn.modules import ES_MOE

    z = load_npz(golden_dir / f"esmoe_{case}.npz")
    m = ES_MOE(64, 64, **ESMOE_MODES[case])
    m.load_state_dict({k: torch.from_numpy(z[f"sd::{k}"]) for k in z["keys"].tolist()})
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps = 1e-3
    if case == "disabled":
        m.enable_sparse_inference(False)
    m.eval().to(DEV)
    with torch.inference_mode():
        y = m(torch.from_numpy(z["x"]).to(DEV))
    r = m.last_route
    assert np.array_equal((r["gate_w"] > 0).cpu().numpy(), z["retained"]), "retained / contributing expert set differs"
    assert np.abs(r["route_w"].cpu().numpy() - z["route_w"]).max() <= 1e-6
    assert np.abs(r["gate_w"].cpu().numpy() - z["gate_w"]).max() <= 1e-6
    assert np.abs(m.expert_usage_counts.cpu().numpy() - z["usage"]).max() <= 1e-6, "expert_usage_counts"
    assert abs(float(m.load_balancing_loss) - float(z["lb_loss"])) <= 1e-5, "load_balancing_loss"
    assert m.get_expert_usage_stats()["expert_usage"] == pytest.approx(z["usage"].tolist(), abs=1e-6)
    assert_close(y.contiguous(), torch.from_numpy(z["y"]), torch.float32, f"ES_MOE[{case}] output")


def test_esmoe_nonfinite_raises():
    from yolo_master_amd import MoERouterError
    from yolo_master_amd.nn.modules import ES_MOE

    m = ES_MOE(32, 32)
    module_sd(m)
    m.eval().to(DEV)
    x = torch.randn(2, 32, 8, 8)
    x[1, 3, 2, 2] = float("nan")
    with pytest.raises(MoERouterError):
        m(x.to(DEV))
    with pytest.raises(MoERouterError):
        m(torch.randn(2, 32, 8).to(DEV))
    from yolo_master_amd import ShapeMismatchError

    with pytest.raises(ShapeMismatchError):
        m(torch.randn(2, 16, 8, 8).to(DEV))


@pytest.mark.parametrize("nc", [80, 3, 1])
@pytest.mark.parametrize("dtype", DTYPES)
def test_detect_head(dtype, nc):
    from oracle import model_ref
    from yolo_master_amd.nn.modules import Detect

    Detect.legacy = False
    m = Detect(nc, 16, False, [64, 128, 128])   # nc = 3, 1: class rows padded to 16 bytes by the tail conv
    m.stride = torch.tensor([8.0, 16.0, 32.0])
    sd = module_sd(m)
    feats = [rnd(2, 64, 16, 20, seed=1), rnd(2, 128, 8, 10, seed=2), rnd(2, 128, 4, 5, seed=3)]
    with torch.inference_mode():
        ref_y, ref_boxes, ref_scores = model_ref.detect(sd, "model.0", [_prep(f, dtype) for f in feats], [8, 16, 32], nc=nc)
    from yolo_master_amd.nn.modules import set_compute_dtype

    m.eval().to(DEV)
    set_compute_dtype(m, dtype)
    with torch.inference_mode():
        y, preds = m([f.to(DEV) for f in feats])
    assert_close(preds["boxes"], ref_boxes, dtype, "Detect raw boxes")
    assert_close(preds["scores"], ref_scores, dtype, "Detect raw scores")
    assert_close(y[:, 4:], ref_y[:, 4:], dtype, "Detect scores")
    assert_close(y[:, :4], ref_y[:, :4], dtype, "Detect decoded boxes")


def test_detect_decode_exact():
    """Decode kernel alone on fp32 logits: same op order as the reference -> tight tolerance."""
    from oracle import model_ref
    from yolo_master_amd import ops

    B, Hl, Wl, nc = 2, 7, 9, 80
    box = rnd(B, Hl, Wl, 64, seed=1, scale=2.0)
    cls = rnd(B, Hl, Wl, nc, seed=2, scale=3.0)
    y = torch.zeros((B, 84, Hl * Wl + 5), device=DEV)
    ops.detect_decode(box.to(DEV), cls.to(DEV), y, 16.0, 5, 16)
    bx = box.reshape(B, -1, 64).permute(0, 2, 1)
    dist = F.conv2d(bx.reshape(B, 4, 16, -1).transpose(2, 1).softmax(1), torch.arange(16.0).view(1, 16, 1, 1)).view(B, 4, -1)
    sx = torch.arange(Wl) + 0.5
    sy = torch.arange(Hl) + 0.5
    gy, gx = torch.meshgrid(sy, sx, indexing="ij")
    anc = torch.stack((gx, gy), -1).view(-1, 2).t().unsqueeze(0)
    lt, rb = dist.chunk(2, 1)
    ref_box = torch.cat([((anc - lt) + (anc + rb)) / 2, (anc + rb) - (anc - lt)], 1) * 16.0
    got = y[:, :, 5:].cpu()
    assert (got[:, :4] - ref_box).abs().max().item() <= 2e-4
    assert (got[:, 4:] - cls.reshape(B, -1, nc).permute(0, 2, 1).sigmoid()).abs().max().item() <= 1e-6
    assert float(y[:, :, :5].abs().max()) == 0.0


@pytest.mark.parametrize("ties", [False, True])
def test_decode_side_outputs_feed_nms(ties):
    """The decode kernel hands NMS every anchor's best class (tests/helpers.decode_best_then_nms): bit-exact with the plain path."""
    from tests.helpers import decode_best_then_nms

    decode_best_then_nms(DEV, levels=((80, 80, 8.0), (40, 40, 16.0), (20, 20, 32.0)), B=3, nc=80, ties=ties, seed=9)


# ------------------------------------------------------------------------------- NMS
def _nms_compare(y, **kw):
    from oracle import nms_ref
    from yolo_master_amd.nms import non_max_suppression

    ref, ref_idx = nms_ref.non_max_suppression(y.numpy(), return_idxs=True, **kw)
    got, got_idx = non_max_suppression(y.to(DEV), return_idxs=True, **kw)
    for b in range(y.shape[0]):
        assert np.array_equal(got_idx[b].cpu().numpy(), ref_idx[b]), f"image {b}: kept anchor indices differ"
        assert np.array_equal(got[b].cpu().numpy(), ref[b]), f"image {b}: detections differ (bit-exact expected)"
    return got


@pytest.mark.parametrize("case", ["single", "multi", "agnostic", "caps", "empty", "one", "classes", "classes_multi",
                                  "seg", "seg_multi", "seg_caps"])      # seg*: `nc` + mask rows, the segment predictor's call
def test_nms_golden(case, golden_dir):
    """HIP NMS == the real reference's non_max_suppression output (fixtures) bit for bit."""
    from tests.helpers import load_npz
    from yolo_master_amd.nms import non_max_suppression

    z = load_npz(golden_dir / f"nms_{case}.npz")
    kw = dict(conf_thres=float(z["arg_conf_thres"]), iou_thres=float(z["arg_iou_thres"]),
              multi_label=bool(z["arg_multi_label"]), agnostic=bool(z["arg_agnostic"]), max_det=int(z["arg_max_det"]),
              max_nms=int(z["arg_max_nms"]))
    if "arg_classes" in z:
        kw["classes"] = z["arg_classes"].tolist()
    if "arg_nc" in z:
        kw["nc"] = int(z["arg_nc"])
    y = torch.from_numpy(z["y"])
    got, idx = non_max_suppression(y.to(DEV), return_idxs=True, **kw)
    for b in range(y.shape[0]):
        assert np.array_equal(idx[b].cpu().numpy(), z[f"idx{b}"]), f"{case} image {b}: kept indices differ"
        assert np.array_equal(got[b].cpu().numpy(), z[f"dets{b}"]), f"{case} image {b}: detections differ"


DENSE_KW = {"val640": dict(conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300),
            "single1280": dict(conf_thres=0.001, iou_thres=0.7, max_det=300)}


@pytest.mark.parametrize("case", list(DENSE_KW))
def test_nms_dense_scene_vs_reference_golden(case, golden_dir):
    """The validator's NMS settings on a dense scene: 672 000 multi-label candidates per image (val640) / 33 600 best-class
    candidates (single1280) against max_nms = 30 000 — the reference sorts them all and truncates (utils/nms.py:142-146); libymk
    selects the same 30 000 by a radix select on the score bits.  Kept anchors and detections bit-exact vs the REAL reference
    (tests/golden/make_golden_nms_dense.py), plus a tie stress against the oracle (quantised scores: hundreds of candidates share
    the threshold score and only the first few in anchor-major order may pass)."""
    from oracle import nms_ref
    from tests.helpers import dense_pred
    from yolo_master_amd.nms import nms_padded, non_max_suppression
    from yolo_master_amd._lib import FLAG_NMS_OVERFLOW

    z = np.load(golden_dir / "nms_dense.npz")
    B, nc, A, seed = [int(v) for v in z[f"{case}::recipe"]]
    y = dense_pred(B, nc, A, seed, frame=float(seed))
    got, idx = non_max_suppression(y.to(DEV), return_idxs=True, **DENSE_KW[case])
    for b in range(B):
        assert np.array_equal(idx[b].cpu().numpy(), z[f"{case}::idx{b}"]), f"{case} image {b}: kept indices differ"
        assert np.array_equal(got[b].cpu().numpy(), z[f"{case}::dets{b}"]), f"{case} image {b}: detections differ"
    status = nms_padded(y.to(DEV), **DENSE_KW[case])[3]
    assert not int(status.item()) & FLAG_NMS_OVERFLOW, "no candidate count may overflow any more"
    yq = y[:1].clone()
    yq[:, 4:] = (yq[:, 4:] * 512).round() / 512          # ~1300 candidates per distinct score
    _nms_compare(yq, **DENSE_KW[case])


def test_nms_random_and_ties():
    g = torch.Generator().manual_seed(21)
    B, nc, A = 4, 80, 8400
    xy = torch.rand(B, 2, A, generator=g) * 600 + 20
    wh = torch.rand(B, 2, A, generator=g) * 200 + 4
    cls = torch.sigmoid(torch.randn(B, nc, A, generator=g) * 1.5 - 4.5)
    cls[1] = (cls[1] * 64).round() / 64            # heavy score ties: stable order must match the oracle
    cls[2] *= 0.0                                   # empty image in the middle of the batch
    cls[3, :, 100:] *= 0.01                         # a handful of candidates
    y = torch.cat([xy, wh, cls], 1)
    _nms_compare(y, conf_thres=0.25, iou_thres=0.7)
    _nms_compare(y, conf_thres=0.25, iou_thres=0.45, agnostic=True, max_det=17)


def test_nms_all_anchors_candidates():
    """Maximum single-label load: every anchor is a candidate (8400 x 8400 IoU mask)."""
    g = torch.Generator().manual_seed(22)
    B, nc, A = 2, 80, 8400
    xy = torch.rand(B, 2, A, generator=g) * 640
    wh = torch.rand(B, 2, A, generator=g) * 60 + 2
    cls = torch.rand(B, nc, A, generator=g) * 0.5 + 0.3
    _nms_compare(torch.cat([xy, wh, cls], 1), conf_thres=0.25, iou_thres=0.7)
    # the same load with 17 distinct scores: the radix index sort (more than 1024 candidates) must keep equal scores in anchor order
    _nms_compare(torch.cat([xy, wh, (cls * 32).round() / 32], 1), conf_thres=0.25, iou_thres=0.7)
    # ... and with scores spread over many binades (no digit is skipped)
    wide = torch.rand(B, nc, A, generator=g) ** 8 * 0.9 + 0.002
    _nms_compare(torch.cat([xy, wh, wide], 1), conf_thres=0.001, iou_thres=0.6, max_det=100)


@pytest.mark.parametrize("agnostic", [False, True])
def test_cw_refine(agnostic):
    """fp64 kernel vs the fp64 oracle (itself pinned to the reference's compiled C++, tests/test_oracle_cw.py): the only
    difference left is the final fp32 rounding of the stored box (< 3e-5 px at these magnitudes) -> 1e-4, the north star's bar."""
    from oracle import nms_ref
    from yolo_master_amd.nms import non_max_suppression

    g = torch.Generator().manual_seed(23)
    B, nc, A = 2, 8, 1500
    xy = torch.rand(B, 2, A, generator=g) * 300 + 20
    wh = torch.rand(B, 2, A, generator=g) * 100 + 20
    cls = torch.sigmoid(torch.randn(B, nc, A, generator=g) * 1.5 - 2.0)
    y = torch.cat([xy, wh, cls], 1)
    plain, idx = non_max_suppression(y.to(DEV), 0.25, 0.6, return_idxs=True, agnostic=agnostic)
    cw = non_max_suppression(y.to(DEV), 0.25, 0.6, cluster=True, sigma=0.1, agnostic=agnostic)
    for b in range(B):
        # candidates exactly as the kernel sees them (single label)
        p = np.transpose(y[b].numpy(), (1, 0)).copy()
        p[:, :4] = nms_ref.xywh2xyxy(p[:, :4])
        conf, j = p[:, 4:].max(1), p[:, 4:].argmax(1)
        m = conf > 0.25
        cands = np.concatenate([p[m, :4], conf[m, None], j[m, None].astype(np.float32)], 1)
        anchor_of = np.nonzero(m)[0]
        keep = [int(np.nonzero(anchor_of == a)[0][0]) for a in idx[b].cpu().numpy()]
        ref = nms_ref.cw_refine(cands, np.array(keep), 0.6, 0.1, agnostic=agnostic)
        got = cw[b].cpu().numpy()
        assert np.array_equal(got[:, 4:], plain[b].cpu().numpy()[:, 4:]), "CW-NMS must not change scores/classes"
        assert np.abs(got[:, :4] - ref).max() <= 1e-4, f"CW-NMS boxes differ: {np.abs(got[:, :4] - ref).max()}"
        assert np.abs(got[:, :4] - plain[b].cpu().numpy()[:, :4]).max() > 1e-3, "refinement had no effect"


def _stem_pair_vs_unfused(case):
    """(fused output, |fused - unfused libymk pair|) for one image-size case."""
    from tests.test_hostemu_stem2 import run_case
    from yolo_master_amd import _lib, ops

    got = run_case(_lib.load(), case, dev="cuda:0", stream=None)
    torch.cuda.synchronize()
    B, H, W = case
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.rand(B, 3, H, W, generator=g)
    w0 = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    b0 = torch.randn(32, generator=g) * 0.3
    w1 = torch.randn(64, 32, 3, 3, generator=g) * (9 * 32) ** -0.5
    b1 = torch.randn(64, generator=g) * 0.2
    wk = w0.permute(0, 2, 3, 1).reshape(32, 27).contiguous().cuda()
    h = ops.conv2d_stem(x.cuda(), wk, b0.cuda(), 3, 2, True, torch.bfloat16, wt=wk.t().contiguous())
    y = ops.conv2d(h, ops.pack_conv_weight(w1, torch.bfloat16).cuda(), b1.cuda(), 3, 2, True)
    return got, (y.float().cpu() - got).abs(), y.float().cpu()


STEM_PAIR_CASES = __import__("tests.test_hostemu_stem2", fromlist=["CASES"]).CASES + [(4, 640, 640), (2, 320, 324)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", STEM_PAIR_CASES)
def test_stem_pair(case):
    """Fused stem + row-1 convolution (csrc/stem2.hip, default = bf16-split stem operands) vs the two-layer composition in torch
    (inside run_case) and vs the unfused libymk pair it replaces: the stem map differs from the fp32-matrix-core stem in about
    0.3 % of its values by one bf16 ulp, which shows up as isolated last-bit differences of row 1's output."""
    got, d, y = _stem_pair_vs_unfused(case)
    scale = max(1.0, float(y.abs().max()))
    assert float(d.max()) <= 2e-2 * scale, f"max |fused - unfused| {float(d.max()):.3e}"
    assert float(d.mean()) <= 2e-4 * scale, f"mean |fused - unfused| {float(d.mean()):.3e}"


@pytest.mark.gpu
def test_stem_pair_fp32_variant_is_bit_identical_to_the_unfused_pair():
    """YMK_DISABLE bit 4096: the fused kernel with the stem on the fp32 matrix cores reproduces ymk_conv2d_stem_nchw -> ymk_conv2d
    bit for bit (own process: the switch is read once per process)."""
    import os
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.test_gpu_kernels import _stem_pair_vs_unfused, STEM_PAIR_CASES\n"
            "for c in STEM_PAIR_CASES:\n"
            "    got, d, y = _stem_pair_vs_unfused(c)\n"
            "    assert float(d.max()) == 0.0, (c, float(d.max()), int((d > 0).sum()))\n"
            "print('IDENTICAL', len(STEM_PAIR_CASES))\n") % str(Path(__file__).resolve().parent.parent)
    env = dict(os.environ, YMK_DISABLE="4096")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env,
                       cwd=str(Path(__file__).resolve().parent.parent))
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("case", __import__("tests.test_hostemu_c3k2f", fromlist=["CASES"]).CASES + [(4, 160, 160), (2, 80, 96)])
def test_c3k2_fused(case):
    """Fused C3k2 block (csrc/c3k2f.hip) vs the four-convolution composition in torch (inside run_case) and vs the unfused libymk
    convolutions it replaces (same stage roundings: differences are isolated bf16 ulps where a stage rounds the other way)."""
    from tests.test_hostemu_c3k2f import operands, run_case
    from yolo_master_amd import _lib, ops

    got = run_case(_lib.load(), case, dev="cuda:0", stream=None)
    torch.cuda.synchronize()
    x, ws, bs, packed = operands(case)
    xd, pk, bd = x.cuda(), [w.cuda() for w in packed], [b.cuda() for b in bs]
    y1 = ops.conv2d(xd, pk[0], bd[0], 1, 1, True)
    b = y1[..., 32:]
    h = ops.conv2d(b, pk[1], bd[1], 3, 1, True)
    m = ops.conv2d(h, pk[2], bd[2], 3, 1, True, residual=b)
    y = ops.conv2d(torch.cat([y1, m], -1), pk[3], bd[3], 1, 1, True).float().cpu()
    d = (y - got).abs()
    scale = max(1.0, float(y.abs().max()))
    assert float(d.max()) <= 4e-2 * scale and float(d.mean()) <= 2e-4 * scale, f"max {float(d.max()):.3e} mean {float(d.mean()):.3e}"
    print(f"fused vs unfused libymk: {int((d > 0).sum())} of {d.numel()} elements differ, max {float(d.max()):.3e}")


@pytest.mark.gpu
def test_esmoe_route_from_the_producers_pooled_sums():
    """ymk_esmoe_route_pooled on the per-tile channel sums c3k2_fused leaves (`out.gap_part`) = ymk_esmoe_route reading the map:
    same retained experts, routing weights equal to fp32 summation-order noise."""
    from tests.test_hostemu_c3k2f import operands
    from yolo_master_amd import ops

    x, ws, bs, packed = operands((6, 40, 64))
    pk, bd = [w.cuda() for w in packed], [b.cuda() for b in bs]
    y = ops.c3k2_fused(x.cuda(), *[(w, b) for w, b in zip(pk, bd)])
    assert y.gap_part.shape == (6, 5 * 2, 128)
    g = torch.Generator().manual_seed(1)
    w1, b1 = (torch.randn(32, 128, generator=g) * 0.5).cuda(), (torch.randn(32, generator=g) * 0.1).cuda()
    w2, b2 = (torch.randn(4, 32, generator=g) * 0.8).cuda(), (torch.randn(4, generator=g) * 0.1).cuda()
    flags = torch.zeros(1, dtype=torch.int32, device="cuda")
    pooled = ops.esmoe_route(y, w1, b1, w2, b2, 2, 0.3, flags)
    plain = ops.esmoe_route(y.clone(), w1, b1, w2, b2, 2, 0.3, flags)      # a copy carries no partials: stage 1 reads the map
    torch.cuda.synchronize()
    assert int(flags.item()) == 0, f"flags {int(flags.item())}"
    dw = float((pooled[0] - plain[0]).abs().max())
    assert dw <= 1e-5, f"routing weights differ by {dw:.3e}"
    assert torch.equal(pooled[2], plain[2]), f"retained experts differ:\n{pooled[2].tolist()}\n{plain[2].tolist()}"
    n = int(pooled[3][-1])                                                   # retained pairs: csr_pair is defined up to csr_off[E]
    assert torch.equal(pooled[3], plain[3]) and torch.equal(pooled[4][:n], plain[4][:n]), "CSR differs"
    assert float((pooled[1] - plain[1]).abs().max()) <= 1e-5
    y.gap_part[0, 0, 5] = float("nan")                                     # a non-finite sum must raise the router's input flag
    ops.esmoe_route(y, w1, b1, w2, b2, 2, 0.3, flags)
    assert int(flags.item()) & 1


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 80, 80, 192, 256), (16, 160, 160, 96, 128), (32, 80, 64, 256, 384)], ids=lambda c: "x".join(map(str, c)))
def test_streaming_1x1_pooled_sums_feed_the_router(shape):
    """ops.conv2d(pool=True) -> ymk_conv1x1_pooled (round 5): the C3k2 tail that produces an ES-MoE layer's input leaves the router's pooled
    sums.  Same stored values as the plain convolution; sums = the stored values summed per 128-pixel tile; the router on the sums retains the
    experts the router on the map retains."""
    from yolo_master_amd import ops

    B, H, W, Cin, Cout = shape
    x = rnd(B, H, W, Cin, seed=21).to(torch.bfloat16).to(DEV)
    wp = ops.pack_conv_weight(rnd(Cout, Cin, 1, 1, seed=22, scale=Cin ** -0.5), torch.bfloat16).to(DEV)
    bias = rnd(Cout, seed=23, scale=0.2).to(DEV)
    plain = ops.conv2d(x, wp, bias, 1, 1, True)
    y = ops.conv2d(x, wp, bias, 1, 1, True, pool=True)
    assert torch.equal(y, plain)
    part = y.gap_part
    assert part.shape == (B, H * W // 128, Cout)
    ref = y.float().reshape(B, H * W // 128, 128, Cout).sum(2)
    assert torch.allclose(part, ref, rtol=1e-5, atol=1e-3), float((part - ref).abs().max())
    # a batch too small for the streaming kernel: plain convolution + ymk_pool_tiles128 — the SAME sums, bit for bit, for the same images
    ys = ops.conv2d(x[:2].contiguous(), wp, bias, 1, 1, True, pool=True)
    assert torch.equal(ys, y[:2]) and torch.equal(ys.gap_part, part[:2]), "tile sums depend on the kernel that produced the map"
    g = torch.Generator().manual_seed(2)
    w1, b1 = (torch.randn(32, Cout, generator=g) * 0.5).to(DEV), (torch.randn(32, generator=g) * 0.1).to(DEV)
    w2, b2 = (torch.randn(4, 32, generator=g) * 0.8).to(DEV), (torch.randn(4, generator=g) * 0.1).to(DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    pooled = ops.esmoe_route(y, w1, b1, w2, b2, 2, 0.3, flags)
    byread = ops.esmoe_route(plain, w1, b1, w2, b2, 2, 0.3, flags)
    torch.cuda.synchronize()
    assert int(flags.item()) == 0
    assert float((pooled[0] - byread[0]).abs().max()) <= 1e-5 and torch.equal(pooled[2], byread[2])


@pytest.mark.gpu
@pytest.mark.parametrize("case", __import__("tests.test_hostemu_detcls", fromlist=["BOX_CASES"]).BOX_CASES + [(4, 80, 80, 8.0), (3, 40, 40, 16.0), (2, 20, 20, 32.0)])
def test_detect_box_tail(case):
    """Box-branch tail with the DFL decode in its epilogue (csrc/elementwise.hip detect_box_tail_kernel) vs convolution + detect_decode."""
    from tests.test_hostemu_detcls import run_box_case
    from yolo_master_amd import ops

    run_box_case(ops, case, dev=DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("case", __import__("tests.test_hostemu_detcls", fromlist=["CASES"]).CASES + [(4, 80, 80, 128, 80), (3, 40, 40, 256, 80), (2, 20, 20, 256, 80)])
def test_detect_cls_fused(case):
    """Fused Detect class branch (csrc/detcls.hip) vs the five-convolution composition in torch (inside run_case) and vs the unfused
    libymk kernels it replaces (same stage roundings; isolated bf16-ulp effects where a stage rounds the other way)."""
    from tests.test_hostemu_detcls import operands, run_case
    from yolo_master_amd import _lib, ops

    got = run_case(_lib.load(), case, dev="cuda:0", stream=None)
    torch.cuda.synchronize()
    x, w, b, packed = operands(case)
    pk, bd = {k: v.cuda() for k, v in packed.items()}, {k: v.cuda() for k, v in b.items()}
    h = ops.dwconv2d(x.cuda(), pk["d1"], bd["d1"], 3, True)
    h = ops.conv2d(h, pk["p1"], bd["p1"], 1, 1, True)
    h = ops.dwconv2d(h, pk["d2"], bd["d2"], 3, True)
    h = ops.conv2d(h, pk["p2"], bd["p2"], 1, 1, True)
    y = ops.conv2d(h, pk["w3"], bd["w3"], 1, 1, False, out_dtype=torch.float32).cpu()
    d = (y - got).abs()
    scale = max(1.0, float(y.abs().max()))
    assert float(d.max()) <= 4e-2 * scale and float(d.mean()) <= 3e-4 * scale, f"max {float(d.max()):.3e} mean {float(d.mean()):.3e}"


def test_nms_iou_threshold_ties():
    """IoU thresholds exactly on / one ulp off a pair's float32 IoU: the greedy pass decides as the reference's `inter / union > thr` does —
    kept sets bit-exact against the oracle."""
    from tests.test_hostemu_round2 import run_nms_threshold_ties
    from yolo_master_amd.nms import non_max_suppression

    run_nms_threshold_ties(lambda y, c, t, **kw: non_max_suppression(y.to(DEV), c, t, **kw))
