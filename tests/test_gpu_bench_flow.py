"""bench.py's own flow on the GPU box: N=1 plain, N=1 under torch.distributed.run, and the N=2 flow (two ranks sharing the
one GPU of the test box over gloo: YMK_BENCH_SHARE_GPU=1 — RCCL refuses two ranks on one device).  What is checked is the
FLOW the driver's scaling run uses (rendezvous, weight broadcast, per-rank HIP graph, result all_gather into pre-allocated
buffers, barrier-bracketed timing, one JSON line from rank 0), not a speed."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
ARGS = ["--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"]


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env=None):
    r = subprocess.run(cmd, cwd=ROOT, env={**os.environ, **(env or {})}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected exactly one JSON line, got {len(lines)}: {r.stdout[-2000:]}"
    return json.loads(lines[0])


def _torchrun(n):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(_port()), "bench.py", "--gpus", str(n), *ARGS]


def test_bench_n1_plain_and_under_torchrun():
    a = _run([sys.executable, "bench.py", *ARGS])
    b = _run(_torchrun(1))
    for d in (a, b):
        assert d["n_gpus"] == 1 and d["config"]["launch"] == "hipGraph" and d["config"]["global_batch"] == 8 and d["value"] > 0
        assert d["roofline"] and d["roofline"]["bound"] in ("hbm", "mfma")
        assert "diagnostic" in d["metric"]          # batch 8 is not the headline configuration and is labelled so


def test_bench_two_ranks_share_the_gpu():
    d = _run(_torchrun(2), env={"YMK_BENCH_SHARE_GPU": "1", "YMK_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert d["config"]["launch"] == "hipGraph", "the rank-local step must be a captured graph at N>1 too"
    assert d["value"] > 0 and d["ms_per_step"] > 0
