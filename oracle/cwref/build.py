"""ORACLE build recipe: compile the reference's C++ decode/NMS source IN PLACE (never copied) into
oracle/_ref/libcwref.so, against the OpenCV geometry shim in this directory.  Needs /root/reference and g++.

    python -m oracle.cwref.build
"""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE.parent / "_ref"
REF = Path(os.environ.get("YMK_REFERENCE", "/root/reference")) / "examples/YOLO-Master-Cross-Platform-Edge-Deployment/cpp"


def available() -> bool:
    return (REF / "src/common.cpp").is_file()


def lib_path() -> Path:
    return OUT / "libcwref.so"


def build(verbose: bool = False) -> Path:
    if not available():
        raise RuntimeError(f"reference C++ sources not found under {REF}")
    import fcntl

    OUT.mkdir(exist_ok=True)
    with open(OUT / ".cwref.lock", "w") as fh:   # one builder at a time (parallel test workers); the library appears atomically
        fcntl.flock(fh, fcntl.LOCK_EX)
        tmp = lib_path().with_suffix(".so.tmp")
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", f"-I{HERE}", f"-I{REF / 'include'}",
               str(REF / "src/common.cpp"), str(HERE / "cwref_api.cpp"), "-o", str(tmp)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        tmp.replace(lib_path())
    return lib_path()


if __name__ == "__main__":
    print(build(verbose=True))
