"""Build libymk.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m yolo_master_amd.build` or `__graft_entry__.build()`.  Objects are cached by
source mtime under yolo_master_amd/csrc/_obj/; the shared library lands next to this file
so that it travels with the tree (gpurun snapshot) and is the one the tests load.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = CSRC / "_obj"
LIB = HERE / "libymk.so"
# The same sources compiled a second time with the 16-bit element format = IEEE binary16 (-DYMK_H16_F16, csrc/ymk_common.h): same
# C-ABI, loaded side by side by _lib.py for torch.float16 tensors (the reference's `half=True` mode).
VARIANTS = {"bf16": (OBJ, LIB, []), "f16": (CSRC / "_obj_f16", HERE / "libymk_f16.so", ["-DYMK_H16_F16"])}
ARCH = "gfx950"
# files whose arithmetic must not be contracted into FMAs (bit-exact NMS / decode)
NO_CONTRACT = {"nms.hip", "elementwise.hip", "post.hip", "preproc.hip"}
SOURCES = ["capi.hip", "conv.hip", "dwconv.hip", "mlp.hip", "stem2.hip", "c3k2f.hip", "detcls.hip", "esmoe.hip", "attn.hip", "elementwise.hip", "nms.hip",
           "mixture.hip", "mixattn.hip",   # config-5 rows, first implementation (include/ymk_mixture.h)
           "conv_glds.hip", "post.hip", "preproc.hip"]   # LDS-DMA tiled convolution core, box rescaling / validation matching, letterbox (include/ymk_next.h)
HEADERS = ["ymk_common.h", "igemm.h", "glds.h", "../../include/ymk.h", "../../include/ymk_mixture.h", "../../include/ymk_next.h"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libymk cannot be built (ROCm toolchain required)")


def source_hash() -> str:
    """sha256 (16 hex digits) over the kernel sources and shared headers: committed rocprofv3 summaries carry it, so that a
    profile is only ever quoted for the kernels it was collected on (bench.py `roofline.traffic`, `families[].sq`)."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(SOURCES) + ["ymk_common.h", "igemm.h", "glds.h"]:
        h.update(f.encode())
        h.update((CSRC / f).read_bytes())
    return h.hexdigest()[:16]


def _compile_cmd(hipcc: str, src: str, obj: Path, defs: list) -> list:
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *defs, "-c", str(CSRC / src), "-o", str(obj)]
    if src in NO_CONTRACT:
        cmd.insert(4, "-ffp-contract=off")
    if src in ("post.hip", "preproc.hip"):   # bit-exact box rescaling divides by the gain: IEEE division (hipcc's default, stated explicitly)
        cmd.insert(4, "-fhip-fp32-correctly-rounded-divide-sqrt")
    return cmd


def build(force: bool = False, verbose: bool = True, variants=("bf16", "f16")) -> Path:
    """Compile what is stale (objects are cached by mtime against their source and the shared headers) for every variant, up to
    $YMK_BUILD_JOBS (default: all cores) hipcc processes at a time, then link each variant's shared object."""
    from concurrent.futures import ThreadPoolExecutor

    hipcc = _hipcc()
    hdr_m = max((CSRC / h).resolve().stat().st_mtime for h in HEADERS)
    jobs, relink = [], set()
    for v in variants:
        obj_dir, lib, defs = VARIANTS[v]
        obj_dir.mkdir(exist_ok=True)
        for src in SOURCES:
            o = obj_dir / src.replace(".hip", ".o")
            if not force and o.exists() and o.stat().st_mtime >= max((CSRC / src).stat().st_mtime, hdr_m):
                continue
            jobs.append(_compile_cmd(hipcc, src, o, defs))
            relink.add(v)
        if force or not lib.exists():
            relink.add(v)

    def run(cmd):
        if verbose:
            print("[ymk build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=int(os.environ.get("YMK_BUILD_JOBS", os.cpu_count() or 4))) as ex:
            list(ex.map(run, jobs))
    for v in variants:
        obj_dir, lib, _ = VARIANTS[v]
        if v in relink:
            # -Bsymbolic + hidden-by-default template instantiations are not needed: ctypes loads each library RTLD_LOCAL, so the two
            # variants' identically named symbols never meet
            run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", str(lib), *[str(obj_dir / s.replace(".hip", ".o")) for s in SOURCES]])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
