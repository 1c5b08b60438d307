"""The whole v0 detector on the CPU lane emulator: the product's host code (yolo_master_amd/nn, ops, nms) drives the UNMODIFIED
kernel sources of libymk compiled for the host (tests/hostemu) — every lane a fiber, matrix cores, LDS-DMA, shuffles emulated — and
the result is held against the REAL reference's golden vectors (tests/golden/fwd_n_tiny.npz: N scale, 1 x 3 x 64 x 64, fp32) with the
same tolerance model as the GPU test (tests/test_gpu_model.py).  It exercises on a CPU what otherwise only an MI355X does: kernel
dispatch thresholds, workspace plumbing, the ES-MoE router -> CSR -> depthwise -> pointwise chain, attention, Detect decode and NMS."""
import numpy as np
import pytest
import torch

from tests.helpers import load_npz


@pytest.fixture
def host_model(hostlib, monkeypatch):
    from yolo_master_amd import nms, ops, postprocess

    for mod in (ops, postprocess):
        monkeypatch.setattr(mod, "lib", hostlib)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    return ops


def test_n_model_64px_against_the_reference_fixture(host_model, golden_dir):
    from yolo_master_amd.nms import non_max_suppression
    from yolo_master_amd.nn.tasks import DetectionModel
    from yolo_master_amd.weights import synth_input, synth_state_dict

    z = load_npz(golden_dir / "fwd_n_tiny.npz")
    B, H, W, seed = int(z["B"]), int(z["H"]), int(z["W"]), int(z["seed"])
    m = DetectionModel("yolo-master-n.yaml")
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
    m = m.eval().set_compute_dtype(torch.float32)
    x = synth_input(B, H, W, seed=seed)
    taps = {}
    with torch.inference_mode():
        y, preds = m._predict_once(x, taps=taps)
    m.check_flags()
    for i in (3, 6, 9, 12):   # routing decisions: identical to the reference
        r = m.model[i].last_route
        assert np.array_equal((r["gate_w"] > 0).numpy(), z[f"route{i}_retained"]), f"layer {i}: retained experts differ"
        assert np.abs(r["route_w"].numpy() - z[f"route{i}_route_w"]).max() <= 1e-4
    for i in range(25):       # every layer output, sampled, against the fp64 evaluation of the reference graph
        t = taps[i]
        if not torch.is_tensor(t):
            t = t.materialise()
        got = host_model.nhwc_to_nchw_f32(t)
        assert tuple(got.shape) == tuple(z[f"layer{i}_shape"]), f"layer {i} shape"
        g = got.reshape(-1)[torch.from_numpy(z[f"layer{i}_idx"].astype(np.int64))].double().numpy()
        e64 = float(np.abs(g - z[f"layer{i}_val64"]).max())
        scale = float(np.abs(z[f"layer{i}_val64"]).max())
        bound = 3.0 * float(z[f"layer{i}_noise"]) + 1e-4 * max(scale, 1.0)
        assert e64 <= bound, f"layer {i}: |emulated hip - fp64| = {e64:.3e} > {bound:.3e}"
    assert tuple(y.shape) == tuple(z["y_shape"])
    g = y.reshape(-1)[torch.from_numpy(z["y_idx"].astype(np.int64))].double().numpy()
    is_box = (z["y_idx"].astype(np.int64) // y.shape[2]) % y.shape[1] < 4
    eb = float(np.abs(g - z["y_val64"])[is_box].max())
    ec = float(np.abs(g - z["y_val64"])[~is_box].max())
    assert eb <= 3.0 * float(z["y_noise_box"]) + 1e-4 + 1e-4 * float(np.abs(z["y_val64"][is_box]).max()), f"boxes {eb:.3e}"
    assert ec <= 3.0 * float(z["y_noise_cls"]) + 1e-4, f"scores {ec:.3e}"
    dets = non_max_suppression(y, 0.25, 0.7)
    assert len(dets) == B and all(d.shape[1] == 6 for d in dets)


def test_s_model_bf16_64px_against_the_oracle(host_model):
    """The bf16 S detector (the bench configuration's kernels: fused stem pair, fused C3k2 row, table-driven ES-MoE pointwise stage, LDS-DMA
    convolutions, resident area attention, fused MLP and Detect class branch) on the emulator against the fp32 oracle on the same weights
    and input: routing decisions identical, scores and boxes within the bf16 drift measured on the GPU for this depth (4e-4 / 0.3 px here;
    bars at ~4x)."""
    from oracle import model_ref
    from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load
    from yolo_master_amd.weights import synth_input, synth_state_dict

    sd = synth_state_dict(DetectionModel("yolo-master-s.yaml").state_dict(), seed=0)
    m = DetectionModel("yolo-master-s.yaml")
    m.load_state_dict(sd)
    m = m.eval().set_compute_dtype(torch.bfloat16)
    x = synth_input(2, 64, 64, seed=3)
    info = {}
    with torch.inference_mode():
        oy, _, _ = model_ref.forward(yaml_model_load("yolo-master-s.yaml"), sd, x, moe_info=info)
        y, _ = m._predict_once(x)
    m.check_flags()
    for i in (3, 6, 9, 12):
        assert torch.equal(m.model[i].last_route["gate_w"] > 0, info[f"model.{i}"]["retained"]), f"layer {i}: retained experts differ"
    err = (y.float() - oy).abs()
    assert float(err[:, 4:].max()) <= 2e-3, f"scores {float(err[:, 4:].max()):.3e}"
    assert float(err[:, :4].max()) <= 1.0, f"boxes {float(err[:, :4].max()):.3e} px"


@pytest.fixture(scope="module")
def hostlib_f16():
    """libymk_hostemu_f16.so: the fp16 build of the kernel sources (-DYMK_H16_F16: IEEE binary16 elements, v_mfma_f32_16x16x32_f16)
    compiled for the host, bound with the product's ctypes tables."""
    import ctypes as C

    from tests.hostemu import build as hostemu_build
    from yolo_master_amd import _lib

    path = hostemu_build.build(f16=True)
    if path is None:
        pytest.skip("no host clang++ to build the kernel emulation")
    h = C.CDLL(str(path))
    for name, (res, args) in {**_lib.SYMBOLS, **_lib.SYMBOLS_MIXTURE, **_lib.SYMBOLS_NEXT}.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
    assert h.ymk_h16_format() == _lib.H16_FORMAT_F16
    return h


def test_s_model_fp16_64px_against_the_oracle(hostlib_f16, monkeypatch):
    """The fp16 build (the reference's `half=True` precision, engine/predictor.py:174,415): the SAME kernel sources with the 16-bit
    element format switched to IEEE binary16, S detector through every fused kernel of the bench configuration, on the emulator
    against the fp32 oracle: routing identical; with 3 more mantissa bits than bf16 the drift must be several times smaller than the
    bf16 test's bars (2e-3 / 1 px there)."""
    from oracle import model_ref
    from yolo_master_amd import ops, postprocess
    from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load
    from yolo_master_amd.weights import synth_input, synth_state_dict

    for mod in (ops, postprocess):
        monkeypatch.setattr(mod, "lib", hostlib_f16)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    monkeypatch.setattr(ops, "HAS_F16", True)
    sd = synth_state_dict(DetectionModel("yolo-master-s.yaml").state_dict(), seed=0)
    m = DetectionModel("yolo-master-s.yaml")
    m.load_state_dict(sd)
    m = m.eval().set_compute_dtype(torch.float16)
    x = synth_input(2, 64, 64, seed=3)
    info = {}
    with torch.inference_mode():
        oy, _, _ = model_ref.forward(yaml_model_load("yolo-master-s.yaml"), sd, x, moe_info=info)
        y, _ = m._predict_once(x)
    m.check_flags()
    for i in (3, 6, 9, 12):
        assert torch.equal(m.model[i].last_route["gate_w"] > 0, info[f"model.{i}"]["retained"]), f"layer {i}: retained experts differ"
    err = (y.float() - oy).abs()
    print(f"fp16 S detector on the emulator: scores {float(err[:, 4:].max()):.3e}, boxes {float(err[:, :4].max()):.3e} px")
    assert float(err[:, 4:].max()) <= 5e-4, f"scores {float(err[:, 4:].max()):.3e}"
    assert float(err[:, :4].max()) <= 0.25, f"boxes {float(err[:, :4].max()):.3e} px"
