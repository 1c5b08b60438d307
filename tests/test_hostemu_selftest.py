"""The CPU lane emulator's own primitives (tests/hostemu/hip/hip_runtime.h) against what the hardware does: wave shuffle
addressing, workgroup barriers around LDS traffic, the 16x16x32 bf16 matrix-core operand / result layout, LDS-DMA lane
placement.  The layouts are the ones the library's validated kernels rely on (tools/micro/gemm256.hip ran on MI355X with the
same fragment code), so a change of the emulator that breaks them is caught here rather than in a kernel test."""
import subprocess

import pytest

from tests.hostemu import build as hostemu_build


def test_emulator_primitives():
    exe = hostemu_build.build_selftest()
    if exe is None:
        pytest.skip("no host clang++ to build the kernel emulation")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SELFTEST OK" in r.stdout, r.stdout + r.stderr
