"""The config-5 HIP kernels' LOGIC, checked on the CPU before they ever reach an MI355X.

`tests/hostemu/` compiles csrc/mixture.hip and csrc/mixattn.hip — the unmodified kernel sources — for the host with a
stand-in <hip/hip_runtime.h> that runs every GPU lane as a fiber (workgroup barriers, wave shuffles, LDS as
function-static storage).  The product's own wrappers (`yolo_master_amd.ops`, YMK_EXPERIMENTAL=1) then call that library
through the same ctypes tables and the same C-ABI as libymk, and the bodies of `tests/test_gpu_mixture.py` run unchanged
with DEV = "cpu": every entry point against its contract restatement in fp32 and bf16, the 18 module fixtures of the
real reference and the whole config-5 detector.  What this cannot see is anything hardware-specific (LDS capacity,
launch limits, wave-level timing): that is what the GPU run of the same tests is for."""
import warnings

import pytest
import torch

from tests import emu_ops
from tests.hostemu import build as hostemu_build

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kernels_norms_and_elementwise(T, dtype):
    T.test_norms_and_elementwise(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kernels_pools_stats_shuffle_gather(T, dtype):
    T.test_pools_stats_shuffle_gather(dtype)


def test_kernels_router_tails(T):
    T.test_router_tails()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kernels_attention_family(T, dtype):
    T.test_attention_family(dtype)


def _module_cases():
    import tests.test_gpu_mixture as gpu_tests

    mark = [m for m in gpu_tests.test_modules_vs_reference_golden.pytestmark if m.name == "parametrize"][0]
    return mark.args[1]


@pytest.mark.parametrize("fam,name,ctor", _module_cases())
def test_kernels_modules_vs_reference_golden(T, fam, name, ctor, golden_dir):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        T.test_modules_vs_reference_golden(fam, name, ctor, golden_dir)


@pytest.mark.parametrize("fam,name,ctor", [
    ("moa", "exact", ("MoABlock", (48,), dict(num_heads=6))), ("mot", "dense", ("MoTBlock", (48,), dict(num_heads=6, top_k=3))),
    ("mot", "shift", ("MoTBlock", (48,), dict(num_heads=6, window_shift=True, local_attn_window=7))),
])
def test_kernels_modules_bf16_vs_reference_golden(T, fam, name, ctor, golden_dir):
    """The 16-bit module forms (folded layer scales, matrix-core attention) through the host-compiled kernels."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        T.test_modules_16bit_vs_reference_golden(fam, name, ctor, torch.bfloat16, golden_dir)


@pytest.mark.parametrize("tag", ["cfg5", "v15", "v04", "v01", "v03", "v08s", "uomoe"])
def test_kernels_config5_model_vs_reference_golden(T, tag, golden_dir):
    T.test_config5_model_vs_reference_golden(tag, golden_dir)


@pytest.mark.parametrize("name", ["div_base", "div_e16", "mh_base", "opt_base"])
def test_kernels_gated_v12_to_v15(T, name, golden_dir):
    """OptimalHybridGateMoE / MultiHeadRouterMoE / DiversifiedExpertMoE (per-expert dilated depthwise: ymk_expert_dw3) through the
    host-compiled kernels against the real reference's vectors."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        T.test_gated_v12_v15_vs_reference_golden(name, golden_dir)


def test_conv256_probe_on_the_emulator():
    """The next tiled-GEMM core's design probe (tools/micro/conv256.hip: 256-pixel tiles, LDS-DMA staging with source-side
    swizzle, zero page for border taps, 2- and 3-stage k-loops, XCD tile order) with the matrix core, the LDS-DMA and the
    barriers emulated: all six variants reproduce the naive convolution on shapes with borders, tail tiles, 1x1 and
    several cout tiles per pixel tile.  (DMA completes at once on the emulator, so the counted-vmcnt schedule itself is
    only checked on hardware.)"""
    import subprocess

    exe = hostemu_build.build_probe("conv256")
    if exe is None:
        pytest.skip("no host clang++ to build the kernel emulation")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert "RESULT: all variants match the naive convolution" in r.stdout
    assert r.stdout.count(" OK") == 21 and "MISMATCH" not in r.stdout


def test_kernels_config5_L_scale_model(T):
    """BASELINE config 5's own shape (L scale: MoA heads of 21 -> 24 channels, random-feature basis with nb != padded head
    width, 16-expert shared-inverted block, 512-channel maps) through the emulated kernels, layer by layer against the oracle."""
    import copy

    import yaml

    from oracle import model_ref
    from tests.helpers import fill_by_name
    from yolo_master_amd.nn.tasks import CFG_DIR, DetectionModel

    d = yaml.safe_load(open(CFG_DIR / "yolo-master-moa-mot.yaml"))
    d["scales"]["l"] = [1.0, 1.0, 512]
    d["scale"] = "l"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DetectionModel(copy.deepcopy(d))
    spec = {k: list(v.shape) for k, v in m.state_dict().items() if v.is_floating_point() and v.dim() > 0 and not k.endswith("_rf_matrix")}
    full = dict(m.state_dict())
    full.update(fill_by_name(spec, seed=9, gain=0.7))
    m.load_state_dict(full)
    m.eval()
    x = torch.rand(1, 3, 160, 128, generator=torch.Generator().manual_seed(1))
    taps, otaps = {}, {}
    with torch.inference_mode():
        y, _ = m._predict_once(x, taps=taps)
        oy, _, _ = model_ref.forward(copy.deepcopy(d), dict(m.state_dict()), x, fused=False, taps=otaps)
    for i in range(len(m.model) - 1):
        t = taps[i] if torch.is_tensor(taps[i]) else taps[i].materialise()
        err = float((emu_ops.nhwc_to_nchw_f32(t) - otaps[i]).abs().max() / max(1.0, float(otaps[i].abs().max())))
        assert err <= 1e-4, f"layer {i} ({type(m.model[i]).__name__}): scaled max error {err:.3e}"
    assert float((y[:, 4:] - oy[:, 4:]).abs().max()) <= 1e-5
    m.set_compute_dtype(torch.bfloat16)             # bf16 kernels at these widths: finite, and the vector paths' alignment holds
    with torch.inference_mode():
        yb, _ = m._predict_once(x)
    assert bool(torch.isfinite(yb).all())


def test_kernels_sparse_expert_dispatch_in_the_L_scale_gated_blocks(T, hostlib, monkeypatch):
    """The routed experts of the gated blocks run through ymk_expert_conv_glds (only the routed filter banks); options.OPTIONS.expert_conv_glds = False (YMK_DISABLE bit 512)
    puts the all-experts convolution + gather back.  bf16, L-scale widths (bottleneck 128 -> 4 / 8 banks of 256 couts, 3x3;
    512 -> 16 banks, 1x1): both paths give the same block output up to bf16 rounding of the intermediate."""
    from yolo_master_amd.nn.mixture import VisualEnhancedAdaptiveGateMoE

    for E in (4, 16):
        torch.manual_seed(E)
        m = VisualEnhancedAdaptiveGateMoE(512, 512, num_experts=E, top_k=2).eval()
        m.ymk_dtype = torch.bfloat16
        x = torch.randn(2, 512, 10, 12)
        outs = []
        from yolo_master_amd.options import OPTIONS
        for off in ("512", "0"):
            monkeypatch.setattr(OPTIONS, "expert_conv_glds", off == "0")     # (options.py: YMK_DISABLE bit 512 at import; tests set the field)
            before = emu_ops.CALLS.get("conv2d", 0)
            with torch.inference_mode():
                outs.append(m(x).float())
            calls = emu_ops.CALLS.get("conv2d", 0) - before
            outs.append(calls)
        (dense, n_dense, sparse, n_sparse) = outs
        assert n_sparse == n_dense - 1, "the all-experts convolution must be gone on the sparse path"
        assert float((dense - sparse).abs().max()) <= 3e-2 * max(1.0, float(dense.abs().max()))
