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
    r = subprocess.run(cmd, cwd=ROOT, env={**os.environ, **(env or {})}, capture_output=True, text=True, timeout=240)
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
        assert d["n_gpus"] == 1 and d["config"]["launch"].startswith("hipGraph") and d["config"]["global_batch"] == 8 and d["value"] > 0
        assert d["roofline"] and d["roofline"]["bound"] in ("hbm", "mfma")
        assert "diagnostic" in d["metric"]          # batch 8 is not the headline configuration and is labelled so


def test_bench_reports_the_synchronised_convention_beside_the_pipelined_rate():
    """The driver-run line alone must answer "what is the latency of one batch": `value` (K steps between the barriers, `pipeline_depth`
    batches in flight) and `value_sync` (one batch at a time, synchronised wall-clock, bs * 1000 / p50 — benchmarks/suite.py:316-330),
    forward-only beside forward + NMS, op calls and routed pairs per step."""
    d = _run([sys.executable, "bench.py", *ARGS])
    assert d["pipeline_depth"] == 3 and d["value_sync"] > 0 and d["p50_batch_ms_sync"] > 0
    assert abs(d["value_sync"] - 8 * 1e3 / d["p50_batch_ms_sync"]) <= 0.01 * d["value_sync"]
    assert d["forward_only_sync"]["p50_batch_ms"] <= d["p50_batch_ms_sync"] * 1.05, "forward alone cannot take longer than forward + NMS"
    assert d["p50_batch_latency_ms"] >= d["ms_per_step"] * 0.9 and "p50_ms_per_image" not in d
    assert d["op_calls_per_step"] > 50 and len(d["retained_pairs"]) == 4 and all(8 <= v <= 16 for v in d["retained_pairs"].values())
    assert "sub-batches" in d["config"]["sync_launch"] and d["config"]["nms"].startswith("NMS conf 0.25")
    # round 6: the driver's record keeps `config` and `roofline` and drops unknown top-level keys — the convention numbers live there too
    c, r = d["config"], d["roofline"]
    assert c["value_sync"] == d["value_sync"] and c["p50_batch_ms_sync"] == d["p50_batch_ms_sync"] and c["pipeline_depth"] == 3
    assert c["forward_only_sync"] == d["forward_only_sync"] and r["value_sync"] == d["value_sync"]
    # roofline.frac = achieved / peak of the SCOPE it names (the dominant kernel's layer group where there is one), kernel_frac = the kernel alone
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 2e-4 and abs(r["kernel_frac"] - r["kernel_achieved"] / r["kernel_peak"]) <= 2e-4
    assert r["scope"] and 0 < r["frac"] <= 1 and 0 < r["kernel_frac"] <= 1
    assert set(r["step"]) >= {"hbm_frac", "mfma_frac", "hbm_frac_sync", "mfma_frac_sync"}


def test_bench_rccl_code_path_at_world_size_one():
    """bench.py --force-dist: the `nccl` (RCCL) process group at WORLD_SIZE = 1 and the N > 1 code path on it — weight broadcast, graph
    capture with thread-local error mode next to the process group's watchdog thread, three slots in flight, ONE packed
    all_gather_into_tensor per step on the launching stream.  The 8-GPU scaling run is the driver's; this is the same code on one GPU."""
    d = _run([sys.executable, "bench.py", "--force-dist", *ARGS])
    assert d["n_gpus"] == 1 and d["config"]["collectives"].startswith("RCCL") and "world size 1" in d["config"]["collectives"]
    assert d["config"]["launch"].startswith("hipGraph") and "3 batches in flight" in d["config"]["launch"]
    assert d["value"] > 0 and d["value_sync"] > 0
    e = _run(_torchrun(1) + ["--force-dist"])          # ... and with torchrun's own rendezvous
    assert e["config"]["collectives"].startswith("RCCL") and e["value"] > 0


def test_bench_rccl_code_path_survives_many_replays():
    """The same branch over MANY steps (>= 200 timed + the spin-up replays, three slots in flight, one all_gather per step): the process
    group's watchdog thread passes over the work list hundreds of times while graphs replay and collectives retire — the regime of the
    driver's 8-GPU run that a 3-step test never reaches.  Warm-up collectives are drained by completion (dist.drain_collectives), not by a
    sleep; the line must report how much host time a step costs (`host_us_per_step_launch`: 3 graph replays + 1 gather amortised)."""
    d = _run([sys.executable, "bench.py", "--force-dist", "--steps", "240", "--warmup", "6", "--batch", "8", "--pipeline", "3", "--no-cpu-baseline",
              "--no-sync-leg"])
    assert d["steps"] == 240 and d["config"]["collectives"].startswith("RCCL") and "3 batches in flight" in d["config"]["launch"]
    assert d["value"] > 0 and d["host_us_per_step_launch"] is not None and 0 < d["host_us_per_step_launch"] < 1500


def test_bench_config5_flags():
    """BASELINE config 5 as stated — fp16, CW-NMS, dense-scene NMS settings, expert imbalance — at a size the test box runs in seconds."""
    d = _run([sys.executable, "bench.py", "--cfg", "yolo-master-moa-mot.yaml", "--scale", "n", "--imgsz", "320", "--batch", "2", "--dtype", "f16",
              "--cluster", "--sigma", "0.1", "--dense", "--imbalance", "8,3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["dtype"] == "f16" and "CW-NMS sigma 0.1" in d["config"]["nms"] and "multi_label" in d["config"]["nms"]
    assert d["config"]["imbalance"] == "8,3" and d["value"] > 0 and d["value_sync"] > 0


def test_bench_two_ranks_share_the_gpu():
    d = _run(_torchrun(2), env={"YMK_BENCH_SHARE_GPU": "1", "YMK_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert d["config"]["launch"].startswith("hipGraph"), "the rank-local step must be a captured graph at N>1 too"
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_sub_batches_on_parallel_streams_equal_the_same_sub_batches_in_sequence():
    """bench.py --split S walks the batch as S sub-batches on parallel streams inside the captured graph.  What can go wrong is shared
    mutable scratch between two concurrent walks of ONE model (module instances must not share scratch across streams): the
    concurrent walk must produce bit for bit what the same two sub-batches produce one after the other on one stream."""
    import torch

    sys.path.insert(0, str(ROOT))
    from yolo_master_amd import ops
    from yolo_master_amd.nms import nms_padded
    from yolo_master_amd.nn.tasks import DetectionModel
    from yolo_master_amd.weights import synth_input, synth_state_dict

    dev = torch.device("cuda", 0)
    m = DetectionModel("yolo-master-s.yaml")
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
    m.eval().to(dev).set_compute_dtype(torch.bfloat16)
    x = synth_input(16, 320, 320, seed=5).to(dev)
    words = ops.nms_pack_numel(8, 300)
    with torch.inference_mode():
        seq = torch.empty((2 * words,), dtype=torch.float32, device=dev)
        for i in range(2):
            y, _ = m._predict_once(x[8 * i:8 * i + 8])
            nms_padded(y, 0.25, 0.7, pack=seq[i * words:(i + 1) * words])
        torch.cuda.synchronize()
        for _ in range(3):   # several rounds: a race need not show on the first
            par = torch.empty((2 * words,), dtype=torch.float32, device=dev)
            cur, side = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                yb, _ = m._predict_once(x[8:])
                nms_padded(yb, 0.25, 0.7, pack=par[words:])
            ya, _ = m._predict_once(x[:8])
            nms_padded(ya, 0.25, 0.7, pack=par[:words])
            cur.wait_stream(side)
            torch.cuda.synchronize()
            assert torch.equal(par.view(torch.int32), seq.view(torch.int32)), "concurrent sub-batch walks differ from the sequential ones"
    assert int(ops.nms_pack_views(seq.view(2, words), 8, 300)[1].sum()) > 0


def test_two_captured_steps_in_flight_equal_one_at_a_time():
    """bench.py --pipeline 2: step i + 1 is replayed on a second stream from a second captured graph (own result buffer, own private
    memory pool) while step i still runs — two concurrent walks of ONE model over the SAME resident batch.  Both must produce bit for
    bit what one eager step produces, replay after replay."""
    import torch

    sys.path.insert(0, str(ROOT))
    from yolo_master_amd import ops
    from yolo_master_amd.nms import nms_padded
    from yolo_master_amd.nn.tasks import DetectionModel
    from yolo_master_amd.weights import synth_input, synth_state_dict

    dev = torch.device("cuda", 0)
    m = DetectionModel("yolo-master-s.yaml")
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
    m.eval().to(dev).set_compute_dtype(torch.bfloat16)
    x = synth_input(16, 320, 320, seed=6).to(dev)
    words = ops.nms_pack_numel(16, 300)
    with torch.inference_mode():
        ref = torch.empty((words,), dtype=torch.float32, device=dev)
        for _ in range(2):
            y, _ = m._predict_once(x)
            nms_padded(y, 0.25, 0.7, pack=ref)
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        packs = [torch.empty((words,), dtype=torch.float32, device=dev) for _ in range(2)]
        graphs = []
        for st, pk in zip(streams, packs):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                y, _ = m._predict_once(x)
                nms_padded(y, 0.25, 0.7, pack=pk)
            graphs.append(g)
        torch.cuda.synchronize()
        for _ in range(4):
            for pk in packs:
                pk.zero_()
            torch.cuda.synchronize()
            for st, g in zip(streams, graphs):       # both in flight at once
                with torch.cuda.stream(st):
                    g.replay()
            torch.cuda.synchronize()
            for pk in packs:
                assert torch.equal(pk.view(torch.int32), ref.view(torch.int32)), "a step replayed beside another differs from the eager step"
    m.check_flags()
    assert int(ops.nms_pack_views(ref, 16, 300)[1].sum()) > 0
