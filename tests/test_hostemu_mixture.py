"""The config-5 HIP kernels' LOGIC, checked on the CPU before they ever reach an MI355X.

`tests/hostemu/` compiles csrc/mixture.hip and csrc/mixattn.hip — the unmodified kernel sources — for the host with a
stand-in <hip/hip_runtime.h> that runs every GPU lane as a fiber (workgroup barriers, wave shuffles, LDS as
function-static storage).  The product's own wrappers (`yolo_master_amd.ops`, YMK_EXPERIMENTAL=1) then call that library
through the same ctypes tables and the same C-ABI as libymk, and the bodies of `tests/test_gpu_mixture.py` run unchanged
with DEV = "cpu": every entry point against its contract restatement in fp32 and bf16, the 18 module fixtures of the
real reference and the whole config-5 detector.  What this cannot see is anything hardware-specific (LDS capacity,
launch limits, wave-level timing): that is what the GPU run of the same tests is for."""
import ctypes as C
import warnings

import pytest
import torch

from tests import emu_ops
from tests.hostemu import build as hostemu_build

V0_OPS = ["conv2d", "conv1x1_cat2", "conv2d_stem", "dwconv2d", "dwpw_supported", "dwconv_pwconv", "esmoe_route", "esmoe_dw", "esmoe_pw",
          "esmoe_experts_fused", "area_attn", "upsample2x", "copy_channels", "scale_residual", "nhwc_to_nchw_f32", "detect_decode",
          "nms_batched"]


@pytest.fixture(scope="module")
def hostlib():
    path = hostemu_build.build()
    if path is None:
        pytest.skip("no host clang++ to build the kernel emulation")
    from yolo_master_amd import _lib

    h = C.CDLL(str(path))
    for name, (res, args) in _lib.SYMBOLS_MIXTURE.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


@pytest.fixture
def T(hostlib, monkeypatch):
    """tests/test_gpu_mixture.py re-targeted: mixture entry points -> host-compiled kernels, v0 entry points -> torch
    restatement (they are validated on the GPU already and not under test here), tensors on the CPU."""
    from yolo_master_amd import ops
    import tests.test_gpu_mixture as gpu_tests

    monkeypatch.setenv("YMK_EXPERIMENTAL", "1")
    monkeypatch.setattr(ops, "lib", hostlib)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    for name in V0_OPS:
        monkeypatch.setattr(ops, name, getattr(emu_ops, name))
    monkeypatch.setattr(gpu_tests, "DEV", "cpu")
    return gpu_tests


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


def test_kernels_config5_model_vs_reference_golden(T, golden_dir):
    T.test_config5_model_vs_reference_golden(golden_dir)


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
