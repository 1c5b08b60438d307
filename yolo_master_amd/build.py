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
ARCH = "gfx950"
# files whose arithmetic must not be contracted into FMAs (bit-exact NMS / decode)
NO_CONTRACT = {"nms.hip", "elementwise.hip", "post.hip", "preproc.hip"}
SOURCES = ["capi.hip", "conv.hip", "dwconv.hip", "dwmfma.hip", "mlp.hip", "stem2.hip", "c3k2f.hip", "detcls.hip", "esmoe.hip", "dwpw.hip", "attn.hip", "elementwise.hip", "nms.hip",
           "mixture.hip", "mixattn.hip",   # config-5 rows, first implementation (include/ymk_mixture.h)
           "conv_glds.hip", "post.hip", "preproc.hip"]    # opt-in: next tiled convolution core, box rescaling (include/ymk_next.h)
HEADERS = ["ymk_common.h", "igemm.h", "../../include/ymk.h", "../../include/ymk_mixture.h", "../../include/ymk_next.h"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libymk cannot be built (ROCm toolchain required)")


def build(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    hdr_m = max((CSRC / h).resolve().stat().st_mtime for h in HEADERS)
    objs, rebuilt = [], False
    for src in SOURCES:
        s = CSRC / src
        o = OBJ / (src.replace(".hip", ".o"))
        objs.append(str(o))
        if not force and o.exists() and o.stat().st_mtime >= max(s.stat().st_mtime, hdr_m):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", str(s), "-o", str(o)]
        if src in NO_CONTRACT:
            cmd.insert(4, "-ffp-contract=off")
        if src in ("post.hip", "preproc.hip"):   # bit-exact box rescaling divides by the gain: IEEE division (hipcc's default, stated explicitly)
            cmd.insert(4, "-fhip-fp32-correctly-rounded-divide-sqrt")
        if verbose:
            print("[ymk build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        rebuilt = True
    if rebuilt or force or not LIB.exists():
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *objs]
        if verbose:
            print("[ymk build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
