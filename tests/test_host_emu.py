"""Host-side logic of the product, end to end on the CPU: the real ``DetectionModel`` (YAML parse, weight packing with
BN folding / channel permutations / K padding, NHWC buffer slicing, lazy upsample + virtual concat, ES-MoE routing
bookkeeping, Detect level offsets, NMS front-end) runs with the libymk entry points replaced by ``tests/emu_ops.py``
— a torch restatement of the C-ABI contracts in ``include/ymk.h`` — and is compared with the REAL reference's golden
vectors (``tests/golden/fwd_*.npz``) under the same tolerance model as the GPU parity test.

What this proves: given kernels that honour ``include/ymk.h``, the host code reproduces the reference.  What it does
not prove: the kernels (``-m gpu`` tests do that through the C-ABI).  The emulation lives under ``tests/`` only."""
import numpy as np
import pytest
import torch

from tests.helpers import load_npz


def _model(scale, dtype=torch.float32, cfg=None):
    from yolo_master_amd.nn.tasks import DetectionModel
    from yolo_master_amd.weights import synth_state_dict

    m = DetectionModel(cfg or f"yolo-master-{scale}.yaml")
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
    return m.eval().set_compute_dtype(dtype)


def _sample_err(got_nchw, idx, val):
    g = got_nchw.reshape(-1)[torch.from_numpy(idx.astype(np.int64))].double().numpy()
    return float(np.abs(g - val).max())


@pytest.mark.parametrize("case", ["n_tiny", "n_ragged", "s_small", "l_tiny"])
def test_host_graph_vs_reference_golden(case, golden_dir, emu):
    from yolo_master_amd import ops
    from yolo_master_amd.nms import non_max_suppression
    from yolo_master_amd.weights import synth_input

    z = load_npz(golden_dir / f"fwd_{case}.npz")
    B, H, W, seed = int(z["B"]), int(z["H"]), int(z["W"]), int(z["seed"])
    m = _model(chr(int(z["scale"])))
    x = synth_input(B, H, W, seed=seed)
    taps = {}
    with torch.inference_mode():
        y, preds = m._predict_once(x, taps=taps)
    m.check_flags()
    for i in (3, 6, 9, 12):
        r = m.model[i].last_route
        assert np.array_equal((r["gate_w"] > 0).numpy(), z[f"route{i}_retained"]), f"layer {i}: retained experts differ"
        assert np.abs(r["route_w"].numpy() - z[f"route{i}_route_w"]).max() <= 1e-4
        assert np.abs(r["gate_w"].numpy() - z[f"route{i}_gate_w"]).max() <= 1e-4
    for i in range(25):
        t = taps[i]
        if not torch.is_tensor(t):
            t = t.materialise()
        got = ops.nhwc_to_nchw_f32(t)
        assert tuple(got.shape) == tuple(z[f"layer{i}_shape"]), f"layer {i} shape"
        e64 = _sample_err(got, z[f"layer{i}_idx"], z[f"layer{i}_val64"])
        scale = float(np.abs(z[f"layer{i}_val64"]).max())
        bound = 3.0 * float(z[f"layer{i}_noise"]) + 1e-4 * max(scale, 1.0)
        assert e64 <= bound, f"{case} layer {i}: |host+emu - fp64| = {e64:.3e} > {bound:.3e}"
    assert tuple(y.shape) == tuple(z["y_shape"])
    g = y.reshape(-1)[torch.from_numpy(z["y_idx"].astype(np.int64))].double().numpy()
    is_box = (z["y_idx"].astype(np.int64) // y.shape[2]) % y.shape[1] < 4
    eb = float(np.abs(g - z["y_val64"])[is_box].max())
    ec = float(np.abs(g - z["y_val64"])[~is_box].max())
    assert eb <= 3.0 * float(z["y_noise_box"]) + 1e-4 + 1e-4 * float(np.abs(z["y_val64"][is_box]).max()), f"boxes {eb:.3e}"
    assert ec <= 3.0 * float(z["y_noise_cls"]) + 1e-4, f"scores {ec:.3e}"
    # the NMS front-end (list slicing by counts, index dtype) over the emulated batched kernel
    dets, idx = non_max_suppression(y, float(z["conf"]), float(z["iou"]), return_idxs=True)
    assert len(dets) == B and all(d.shape[1] == 6 for d in dets) and all(i.dtype == torch.int64 for i in idx)
    for b in range(B):
        if bool(z["ties"][b]) or float(z["y_noise_box"]) >= 1e-3:
            continue
        assert np.array_equal(idx[b].numpy(), z[f"nms{b}_idx"]), f"{case} image {b}: kept anchor indices differ"
        assert np.array_equal(dets[b][:, 5].numpy(), z[f"nms{b}_dets"][:, 5])
    # the walk used what it is supposed to use
    assert emu.CALLS["conv2d_stem"] == 1 and emu.CALLS["esmoe_route"] == 4 and emu.CALLS["detect_decode"] == 3
    assert emu.CALLS.get("conv1x1_cat2", 0) >= 1, "neck concatenations are expected to be fused into the consumer's 1x1"


def test_host_graph_bf16_buffers(emu):
    """bf16 compute dtype: every activation buffer the host allocates is bf16, packed weights are bf16 with zero K
    padding (asserted inside the emulation), logits and decode stay fp32."""
    from yolo_master_amd.weights import synth_input

    m = _model("n", torch.bfloat16)
    taps = {}
    with torch.inference_mode():
        y, preds = m._predict_once(synth_input(2, 64, 96, seed=3), taps=taps)
    assert y.dtype == torch.float32 and bool(torch.isfinite(y).all())
    for i, t in taps.items():
        if torch.is_tensor(t) and i != len(m.model) - 1:   # the Detect output y is fp32 by contract
            assert t.dtype == torch.bfloat16, f"layer {i} buffer is {t.dtype}"
    for box, cls in preds["raw"]:
        assert box.dtype == torch.float32 and cls.dtype == torch.float32
    # the reference's eval-mode preds surface (head.py:157-171): {"boxes", "scores", "feats"}, built on access
    assert {"boxes", "scores", "feats"} <= set(preds.keys()) and "boxes" in preds and preds.get("nope") is None
    A = y.shape[2]
    assert tuple(preds["boxes"].shape) == (2, 64, A) and tuple(preds["scores"].shape) == (2, 80, A)
    assert [tuple(f.shape[:2]) for f in preds["feats"]] == [(2, 64), (2, 128), (2, 128)] and len(preds["feats"]) == 3
    assert torch.allclose(preds["scores"].sigmoid(), y[:, 4:], atol=1e-6)
    mf = _model("n", torch.float32)
    with torch.inference_mode():
        yf, _ = mf._predict_once(synth_input(2, 64, 96, seed=3))
    # same network, bf16 storage between layers: scores stay close on the typical anchor
    assert float((y[:, 4:] - yf[:, 4:]).abs().median()) < 5e-3


def test_module_level_api_on_emulation(emu):
    """NCHW-in / NCHW-out module API (what a reference user calls) for the building blocks."""
    from tests.helpers import module_sd
    from oracle import model_ref
    from yolo_master_amd.nn.modules import A2C2f, C3k2, ES_MOE

    torch.manual_seed(0)
    x = torch.randn(2, 64, 16, 24)
    for mod, fn in ((C3k2(64, 64, 2, True), lambda sd: model_ref.c3k2(sd, "model.0", x)),
                    (A2C2f(64, 64, 1, True, 4), lambda sd: model_ref.a2c2f(sd, "model.0", x, 4))):
        sd = module_sd(mod, "model.0", seed=1)
        mod.eval()
        got = mod(x)
        ref = fn(sd)
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), type(mod).__name__
    moe = ES_MOE(64, 64, num_experts=4, top_k=2)
    sd = module_sd(moe, "model.0", seed=2)
    moe.eval()
    got = moe(x)
    info = {}
    ref = model_ref.es_moe(sd, "model.0", x, top_k=2, thr=float(moe.dynamic_threshold), info=info)
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert torch.allclose(moe.expert_usage_counts, info["model.0"]["usage"], atol=1e-6)
    assert torch.allclose(moe.load_balancing_loss, info["model.0"]["lb_loss"], atol=1e-5)
    # dense forward over the router's top-k set (use_sparse_inference off): host logic vs the oracle's dense mode
    moe.enable_sparse_inference(False)
    got = moe(x)
    ref = model_ref.es_moe(sd, "model.0", x, top_k=2, sparse=False)
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_host_graph_fuses_the_stem_pair_at_the_s_width(emu):
    """bf16, S scale (rows 0 / 1 = 3 -> 32 -> 64, both 3x3 stride 2): the graph walk hands both rows to ONE entry point
    (ymk_stem_pair) and produces what the two separate rows produce; with `taps` (per-layer outputs wanted), in fp32, or at
    another width it keeps the two rows."""
    from yolo_master_amd.weights import synth_input

    x = synth_input(1, 64, 96, seed=5)
    m = _model("s", torch.bfloat16)
    with torch.inference_mode():
        emu.CALLS.clear()
        y, _ = m._predict_once(x)
        assert emu.CALLS["stem_pair"] == 1 and emu.CALLS.get("conv2d_stem", 0) == 1   # (the emulation of the pair calls the stem once)
        assert emu.CALLS["c3k2_fused"] == 1                                            # row 2 (64 -> 128, c = 32) as one entry point
        emu.CALLS.clear()
        taps = {}
        yt, _ = m._predict_once(x, taps=taps)
        assert emu.CALLS.get("stem_pair", 0) == 0 and 0 in taps and 1 in taps
    assert torch.equal(y, yt)
    for scale, dt in (("n", torch.bfloat16), ("s", torch.float32)):
        mm = _model(scale, dt)
        with torch.inference_mode():
            emu.CALLS.clear()
            mm._predict_once(x)
        assert emu.CALLS.get("stem_pair", 0) == 0


def test_layer_io_accounting_matches_the_survey_figure(emu):
    """bench.py `roofline_step` / `roofline_layers`: the graph walk reports, per model-YAML layer, the elements the layer reads and writes
    as the REFERENCE's graph defines them (lazy upsamples at their upsampled size, virtual concatenations as the sum of their parts, the
    fused stem pair as two layers).  SURVEY 8(d) measured 32.46 M in + 28.05 M out = 60.5 M elements per 640 x 640 image for
    YOLO-Master-S (121 MB in bf16); ES_MOE rows 3 / 6 / 9 / 12 read and write 5.53 M elements each way."""
    from yolo_master_amd import ops
    from yolo_master_amd.weights import synth_input

    m = _model("s", torch.bfloat16)
    x = synth_input(1, 640, 640, seed=1)
    ops.TIMER.start()
    try:
        ops.TIMER.begin = lambda: None      # (no device events on the CPU: only the walk's bookkeeping is under test)
        with torch.inference_mode():
            m._predict_once(x)
        io = dict(ops.TIMER.io)
    finally:
        del ops.TIMER.begin
        ops.TIMER.stop()
    assert sorted(io) == list(range(len(m.model))), "every YAML layer must be accounted once"
    n_in, n_out = sum(v[0] for v in io.values()), sum(v[1] for v in io.values())
    assert abs(n_in / 32.46e6 - 1) < 0.01 and abs(n_out / 28.05e6 - 1) < 0.01, (n_in, n_out)
    moe = sum(io[i][0] + io[i][1] for i in (3, 6, 9, 12))
    assert moe == 2 * (128 * 160 * 160 + 256 * 80 * 80 + 256 * 40 * 40 + 512 * 20 * 20)
    assert io[8] == (256 * 1600, 256 * 1600) and io[11] == (512 * 400, 512 * 400)
