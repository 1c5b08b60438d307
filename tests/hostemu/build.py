"""TEST INFRASTRUCTURE — compile the config-5 kernel sources for the HOST (tests/hostemu/hip/hip_runtime.h: every GPU lane
a fiber) into tests/hostemu/_build/libymk_hostemu.so, exporting the same `extern "C"` entry points as libymk.

The sources are used as they are, with two textual rewrites the host compiler needs: `extern __shared__ T name[]` -> a pointer
to one global array (dynamic LDS) and `__shared__` -> `static` (workgroups run one after another, so function-static
storage is exactly workgroup-shared storage).  Grid-stride kernels are launched with at most 2 workgroups
(-DYMK_MAX_BLOCKS=2: same code, each lane just walks more elements) to keep the number of fiber set-ups small."""
from __future__ import annotations

import contextlib
import fcntl
import re
import shutil
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "yolo_master_amd" / "csrc"
OUT = HERE / "_build"
SOURCES = ["mixture.hip", "mixattn.hip", "conv_glds.hip", "post.hip", "preproc.hip", "mlp.hip", "stem2.hip", "c3k2f.hip", "detcls.hip",
           "esmoe.hip", "attn.hip", "nms.hip", "conv.hip", "dwconv.hip", "elementwise.hip", "capi.hip"]


@contextlib.contextmanager
def _locked(name: str):
    """One builder at a time per artefact (pytest-xdist workers all ask for the library at start-up): the others wait, then find it up to date."""
    OUT.mkdir(exist_ok=True)
    with open(OUT / f".{name}.lock", "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def compiler():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and Path(c).exists():
            return c
    return None


def build(force: bool = False, f16: bool = False) -> Path | None:
    """f16: the fp16 build of the library (-DYMK_H16_F16: 16-bit element format = IEEE binary16, csrc/ymk_common.h)."""
    cxx = compiler()
    if cxx is None:
        return None
    with _locked("f16" if f16 else "h16"):
        return _build_locked(cxx, force, f16)


def _build_locked(cxx, force: bool, f16: bool) -> Path:
    lib = OUT / ("libymk_hostemu_f16.so" if f16 else "libymk_hostemu.so")
    srcs = [CSRC / s for s in SOURCES]
    deps = srcs + [CSRC / "ymk_common.h", CSRC / "glds.h", CSRC / "igemm.h", ROOT / "include" / "ymk_mixture.h", HERE / "hip" / "hip_runtime.h", Path(__file__)]
    if not force and lib.exists() and lib.stat().st_mtime >= max(d.stat().st_mtime for d in deps):
        return lib
    units = []
    for s in srcs:
        txt = s.read_text()
        txt = re.sub(r"__attribute__\(\(amdgpu_waves_per_eu\([^;{]*?\)\)\)\s*(?=void)", "", txt)   # occupancy hint of GPU kernels
        txt = re.sub(r"\bextern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];", r"\1* \2 = reinterpret_cast<\1*>(hostemu::dyn_lds);", txt)
        txt = re.sub(r"\b__shared__\b", "static", txt)
        txt = txt.replace('#include "ymk_common.h"', f'#include "{CSRC / "ymk_common.h"}"')
        txt = txt.replace('#include "igemm.h"', f'#include "{CSRC / "igemm.h"}"')
        txt = txt.replace('#include "glds.h"', f'#include "{CSRC / "glds.h"}"')
        txt = txt.replace('#include "../../include/ymk_mixture.h"', f'#include "{ROOT / "include" / "ymk_mixture.h"}"')
        u = OUT / (s.stem + ("_host_f16.cpp" if f16 else "_host.cpp"))
        u.write_text(txt)
        units.append(str(u))
    tmp = lib.with_suffix(".so.tmp")   # a loader in another process never sees a half-written library
    cmd = [cxx, "-std=c++20", "-O1", "-fPIC", "-shared", "-pthread", "-Wno-everything", "-DYMK_MAX_BLOCKS=2", "-DGLDS_SMALL_BELOW_DEFAULT=3", "-DNMS_RANK_MAX=1200", "-DYMK_HOST_EMU", *(["-DYMK_H16_F16"] if f16 else []), "-ffp-contract=off", f"-I{HERE}", *units, "-o", str(tmp)]
    subprocess.run(cmd, check=True)
    tmp.replace(lib)
    return lib


def build_probe(name: str = "conv256") -> Path | None:
    """tools/micro/<name>.hip as a host executable (-DYMK_HOST_EMU): the probe's kernel on tiny shapes, self-checked against
    its naive kernel."""
    cxx = compiler()
    if cxx is None:
        return None
    with _locked(f"probe_{name}"):
        return _build_probe_locked(cxx, name)


def _build_probe_locked(cxx, name: str) -> Path:
    src = ROOT / "tools" / "micro" / f"{name}.hip"
    exe = OUT / f"{name}_host"
    deps = [src, HERE / "hip" / "hip_runtime.h", CSRC / "ymk_common.h", Path(__file__)]
    if exe.exists() and exe.stat().st_mtime >= max(d.stat().st_mtime for d in deps):
        return exe
    txt = src.read_text()
    txt = re.sub(r"\bextern\s+__shared__\s+(\w+)\s+(\w+)\[\];", r"\1* \2 = reinterpret_cast<\1*>(hostemu::dyn_lds);", txt)
    txt = re.sub(r"\b__shared__\b", "static", txt)
    txt = txt.replace('"../../yolo_master_amd/csrc/', f'"{CSRC}/')
    u = OUT / f"{name}_host.cpp"
    u.write_text(txt)
    subprocess.run([cxx, "-std=c++20", "-O1", "-Wno-everything", "-DYMK_HOST_EMU", f"-I{HERE}", str(u), "-o", str(exe)], check=True)
    return exe


def build_selftest() -> Path | None:
    cxx = compiler()
    if cxx is None:
        return None
    with _locked("selftest"):
        return _build_selftest_locked(cxx)


def _build_selftest_locked(cxx) -> Path:
    src, exe = HERE / "selftest" / "selftest.hip", OUT / "selftest_host"
    txt = src.read_text()
    txt = re.sub(r"\bextern\s+__shared__\s+(\w+)\s+(\w+)\[\];", r"\1* \2 = reinterpret_cast<\1*>(hostemu::dyn_lds);", txt)
    txt = re.sub(r"\b__shared__\b", "static", txt)
    u = OUT / "selftest_host.cpp"
    u.write_text(txt)
    subprocess.run([cxx, "-std=c++20", "-O1", "-Wno-everything", "-DYMK_HOST_EMU", f"-I{HERE}", str(u), "-o", str(exe)], check=True)
    return exe


if __name__ == "__main__":
    print(build(force=True))
