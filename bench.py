"""Headline benchmark: images/sec @ 640x640, bs=64 per GPU, YOLO-Master-S, forward + batched NMS on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic images already resident in HBM: the 26-layer
forward (libymk HIP kernels) + batched NMS (conf 0.25, IoU 0.7, max_det 300), plus — for N>1 — the RCCL
all_gather of the padded detections.  Images are sharded over ranks (weak scaling: 64 images per GPU, i.e.
BASELINE.json config 3 at N=1 and config 4 at N=8); there is no data-path collective inside the forward.
Rank 0 prints ONE JSON line (schema: see the task contract) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PROFILES = ROOT / "profiles"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E
EAGER_STEPS = 5                # eager per-op-event steps of the roofline leg (per-call medians)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}
TORCH_DTYPE = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


# op family (yolo_master_amd.ops TIMER) -> kernel-name prefix in the rocprofv3 output; single-kernel families only
FAMILY_KERNEL = {
    "moe_pw": "moe_pw_", "moe_dw": "moe_dw_kernel<", "dwconv": "dwconv_kernel<", "area_attn": "area_attn_re", "area_attn_qkv": "area_attn_qkv_kernel<",
    "detect_decode": "detect_decode_kernel", "stem": "stem_",
}


VALU_FAMILIES = ("moe_dw", "dwconv")   # depthwise stencils (csrc/dwconv.hip)
VALU_PEAK_TMACS = 64.0


def _committed_profile(suffix: str):
    """Newest committed rocprofv3 counter summary profiles/*_<suffix>.json whose kernels are the ones in this tree: the summary
    carries the hash of the kernel sources it was collected on (tools/pmc_summary.py, `csrc_sha16`); a profile of other sources is
    STALE and is not used (its numbers would describe kernels that no longer exist)."""
    import glob

    from yolo_master_amd.build import source_hash

    for f in sorted(glob.glob(str(PROFILES / f"*_{suffix}.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("csrc_sha16") == source_hash():
            return d["kernels"]
    return None


def step_counter_traffic():
    """Fabric bytes of ONE eager step (2 x FETCH_SIZE + WRITE_SIZE over every dispatch, tools/micro/step_dispatch_pmc.sh) from the newest
    committed profiles/*_step_dispatch_pmc.txt collected on THIS tree's kernel sources; None otherwise."""
    import glob
    import re

    from yolo_master_amd.build import source_hash

    for f in sorted(glob.glob(str(PROFILES / "*_step_dispatch_pmc.txt")), reverse=True):
        txt = open(f).read()
        m, h = re.search(r"step: fetch ([0-9.]+) GB.*write ([0-9.]+) GB", txt), re.search(r"csrc_sha16 ([0-9a-f]{16})", txt)
        if m and h and h.group(1) == source_hash():
            return {"fetch_gb": float(m.group(1)), "write_gb": float(m.group(2)), "file": os.path.basename(f)}
    return None


def _kernels_of(family: str, table: dict):
    return [family] if family in table else [n for n in table if family in FAMILY_KERNEL and n.startswith(FAMILY_KERNEL[family])]


def pmc_traffic(family: str):
    """HBM bytes per launch of the timed kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_*.json, collected by tools/gpu_profile.sh on the same workload): 2*FETCH_SIZE (gfx950 counts
    wide coalesced reads at half size, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes, averaged over the
    launches.  Convolution records carry the exact kernel name rocprofv3 prints (ops.conv_kernel_name); the other
    op families map to a name prefix.  None when no profile OF THESE SOURCES is committed or the family spans several kernels."""
    fk, wk = _committed_profile("pmc_FETCH_SIZE"), _committed_profile("pmc_WRITE_SIZE")
    if not fk or not wk:
        return None
    tot, n = 0.0, 0
    for name in _kernels_of(family, fk):
        if name in wk:
            c = fk[name]["FETCH_SIZE"]["launches"]
            tot += (2.0 * fk[name]["FETCH_SIZE"]["mean"] + wk[name]["WRITE_SIZE"]["mean"]) * c
            n += c
    return int(tot / n * 1024) if n else None


def sq_utilisation(family: str):
    """Matrix-core / VALU / LDS utilisation of the family's kernel from the committed SQ counter pass (profiles/*_sq_1.json +
    *_sq_2.json, tools/gpu_pmc.sh): the SQ counters of this rocprofv3 sample ONE of the 32 shader engines (SQ_INSTS_MFMA x 32 = the
    kernel's MFMA count), so busy fractions are MFMA_BUSY_CYCLES / (SIMDs of an engine (32) x SQ_BUSY_CYCLES); VALU / LDS figures are
    active-instruction quad-cycles per wave-cycle; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
    s1, s2 = _committed_profile("sq_1"), _committed_profile("sq_2")
    if not s1:
        return None
    acc = {"mfma": 0.0, "busy": 0.0, "valu": 0.0, "lds": 0.0, "wave": 0.0, "conf": 0.0, "ldsact": 0.0, "wait": 0.0, "tot": 0.0}
    for name in _kernels_of(family, s1):
        g = lambda d, c: d.get(name, {}).get(c, {}).get("mean", 0.0) * d.get(name, {}).get(c, {}).get("launches", 0)   # noqa: E731
        acc["mfma"] += g(s1, "SQ_VALU_MFMA_BUSY_CYCLES"); acc["busy"] += g(s1, "SQ_BUSY_CYCLES"); acc["valu"] += g(s1, "SQ_ACTIVE_INST_VALU")
        acc["lds"] += g(s1, "SQ_ACTIVE_INST_LDS"); acc["wave"] += g(s1, "SQ_WAVE_CYCLES")
        if s2:
            acc["conf"] += g(s2, "SQ_LDS_BANK_CONFLICT"); acc["ldsact"] += g(s2, "SQ_LDS_IDX_ACTIVE"); acc["wait"] += g(s2, "SQ_WAIT_ANY")
            acc["tot"] += g(s2, "SQ_WAIT_ANY") + g(s2, "SQ_WAIT_INST_ANY") + g(s2, "SQ_ACTIVE_INST_ANY")
    if not acc["busy"]:
        return None
    out = {"mfma_busy": round(acc["mfma"] / (32.0 * acc["busy"]), 4), "valu_per_wave_cycle": round(acc["valu"] / max(acc["wave"], 1.0), 4),
           "lds_per_wave_cycle": round(acc["lds"] / max(acc["wave"], 1.0), 4)}
    if acc["tot"]:
        out["wave_parked"] = round(acc["wait"] / acc["tot"], 3)
    if acc["ldsact"]:
        out["lds_conflict"] = round(acc["conf"] / acc["ldsact"], 3)
    return out


def cpu_baseline(scale: str, seconds_budget: float = 20.0):
    """Oracle (CPU fp32 restatement of the reference path, kind="port") timed on this host's cores on a
    bounded sample of the same workload: forward + NMS on a few 640x640 images, forward only beside it."""
    from oracle import model_ref, nms_ref
    from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load
    from yolo_master_amd.weights import synth_input, synth_state_dict

    cores = min(os.cpu_count() or 1, 8)    # BASELINE.md section 2 fixes the CPU baseline at 8 threads
    torch.set_num_threads(cores)
    cfg = yaml_model_load(f"yolo-master-{scale}.yaml")
    sd = synth_state_dict(DetectionModel(cfg).state_dict(), seed=0)
    B = 4
    x = synth_input(B, 640, 640, seed=1)
    times, fwd = [], []
    with torch.inference_mode():
        model_ref.forward(cfg, sd, x[:1])  # warm-up (thread pool, oneDNN primitive cache)
        t_all = time.time()
        while len(times) < 5 and (time.time() - t_all < seconds_budget or len(times) < 1):
            t0 = time.time()
            y, _, _ = model_ref.forward(cfg, sd, x)
            t1 = time.time()
            nms_ref.non_max_suppression(y.numpy(), 0.25, 0.7)
            times.append(time.time() - t0)
            fwd.append(t1 - t0)
    out = {"value": round(B / _p50(times), 3), "unit": "images/sec", "cores": cores, "kind": "port",
           "forward_only_value": round(B / _p50(fwd), 3),
           "sample": f"oracle forward+NMS, YOLO-Master-{scale.upper()} fp32, {B}x3x640x640, {len(times)} timed passes (p50)"}
    # the REAL reference timed beside this port (tools/cpu_reference_timing.py): on the GPU box's host when a reference checkout was
    # staged for that call (profiles/*_cpu_reference_gpubox.json), else on the build host (the GPU box has no reference checkout)
    import glob

    # newest round first, whichever of the two names the round used (rNN_reference_on_gpubox.json carries the host-core leg under
    # "reference_cpu_on_gpubox_host"; the older rNN_cpu_reference_gpubox.json is that object alone)
    cands = glob.glob(str(ROOT / "profiles" / "*reference_on_gpubox*.json")) + glob.glob(str(ROOT / "profiles" / "*_cpu_reference*.json"))
    for f in sorted(cands, key=lambda q: os.path.basename(q), reverse=True):
        try:
            ref = json.load(open(f))
            ref = ref.get("reference_cpu_on_gpubox_host", ref)
            if "cores" not in ref:
                continue
            out["reference_timed_beside"] = {"file": os.path.basename(f), "kind": "reference",
                                             **{k: ref[k] for k in ref if k != "same_detection_counts"}}
            break
        except Exception:
            continue
    return out


def _p50(v):
    v = sorted(v)
    return v[len(v) // 2]


def _flush_c_stdio():
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # (no libc handle: nothing buffered there that we could reach)
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scale", default="s")
    ap.add_argument("--cfg", default=None, help="model YAML other than the v0 detector, e.g. yolo-master-moa-mot.yaml with --scale l "
                    "--imgsz 1280 --batch 16 --dtype f16 --cluster --sigma 0.1 --dense --imbalance 8,3 for BASELINE config 5")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"], help="compute type: bf16 (BASELINE config 3), f16 (the reference's "
                    "half=True precision, libymk_f16.so; config 5), f32 (the exact-parity configuration, BASELINE config 2's type)")
    ap.add_argument("--cluster", action="store_true", help="Cluster-Weighted NMS box refinement after the greedy pass (config 5: 'CW-NMS')")
    ap.add_argument("--sigma", type=float, default=0.1, help="CW-NMS kernel width (cfg/default.yaml:196-197)")
    ap.add_argument("--dense", action="store_true", help="dense-scene NMS settings: the validator's conf 0.001 + multi_label (SURVEY 8(d): up to "
                    "max_nms = 30 000 candidates per image reach the ordering / greedy stages)")
    ap.add_argument("--imbalance", default=None, help="expert-imbalance stress A[,T]: expert 0's router logit raised by A in the per-image routers and "
                    "by T (default 3 A / 8) in the per-token routers (SURVEY 8(d) 'imbalance knob'; the config-5 fixture uses 8,3: every image / "
                    "token routes to expert 0)")
    ap.add_argument("--weights", default="cond", choices=["cond", "recipe"], help="cond (default): the PARITY-PINNED state_dict — cfg/cond_<scale>.npz over "
                    "the seeded recipe, the weights tests/test_gpu_baseline_configs.py::test_config3_* hold to the reference's own 16-bit run (same input seed: "
                    "`retained_pairs` here = the fixture's routing); recipe: SURVEY 8(d)'s seeded recipe + BN calibration (rounds 1-4's timed weights, kept for "
                    "round-over-round comparison).  Scales / model families without a cond_<scale>.npz use the recipe")
    ap.add_argument("--roofline-kernel", default=None, help="op family to report in `roofline` (default: the one with the largest share of the step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured HIP graph")
    ap.add_argument("--no-sync-leg", action="store_true", help="skip the synchronised one-batch-at-a-time leg (`value_sync`)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("YMK_BENCH_PIPELINE", "3")), help="batches in flight in the TIMED region: step "
                    "i + 1 (and i + 2) is launched (own stream, own captured graph, own result buffers) while step i is still running — what a "
                    "throughput-oriented server does with several streams; 1 = one step at a time.  `value` is this region's rate; the "
                    "one-batch-at-a-time rate of the reference's own convention is ALWAYS reported beside it as `value_sync`")
    ap.add_argument("--split", type=int, default=int(os.environ.get("YMK_BENCH_SPLIT", "1")), help="walk the batch as this many sub-batches on "
                    "as many HIP streams inside the one captured graph (timed region)")
    ap.add_argument("--sync-split", type=int, default=int(os.environ.get("YMK_BENCH_SYNC_SPLIT", "2")), help="the same for the synchronised leg "
                    "(one batch in flight: the latency-bound launches of the small maps of one sub-batch overlap the other's)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group and run the N > 1 code path (weight broadcast, "
                    "thread-local graph capture, one packed all_gather per step on the launching stream) at WORLD_SIZE = 1: the multi-GPU path's "
                    "smoke test on a 1-GPU box")
    a = ap.parse_args()
    # stdout carries ONE line, the JSON at the end: whatever a library prints through file descriptor 1 in between (RCCL's version banner,
    # MIOpen / hipBLASLt chatter) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from yolo_master_amd import ops
    from yolo_master_amd.dist import broadcast_state_dict, gather_packed, init_from_env
    from yolo_master_amd.nms import nms_padded
    from yolo_master_amd.nn.tasks import DetectionModel
    from yolo_master_amd.weights import expert_imbalance, synth_input, synth_state_dict

    if a.force_dist:
        os.environ["YMK_DIST_FORCE"] = "1"
    rank, local, world = init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    distributed = world > 1 or (a.force_dist and dist.is_initialized())
    if os.environ.get("YMK_BENCH_SHARE_GPU"):  # functional test of the N>1 flow on a 1-GPU box (gloo, all ranks on cuda:0)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = TORCH_DTYPE[a.dtype]

    if a.cfg is None:
        model = DetectionModel(f"yolo-master-{a.scale}.yaml")
    else:   # another model family at a scale its YAML may not list (config 5 = the moa-mot YAML at the L scale)
        from yolo_master_amd.nn.tasks import yaml_model_load

        cfg = yaml_model_load(a.cfg)
        v0 = yaml_model_load("yolo-master.yaml")["scales"]
        cfg.setdefault("scales", {}).setdefault(a.scale, v0[a.scale])
        cfg["scale"] = a.scale
        model = DetectionModel(cfg)
    imb = None
    from yolo_master_amd.weights import CFG_DIR
    cond_file = CFG_DIR / f"cond_{a.scale}.npz"
    use_cond = a.weights == "cond" and a.cfg is None and cond_file.exists()
    if rank == 0:
        sd = synth_state_dict(model.state_dict(), seed=0, calib=str(cond_file)) if use_cond else synth_state_dict(model.state_dict(), seed=0)
        if a.imbalance:
            v = [float(t) for t in a.imbalance.split(",")]
            imb = (v[0], v[1] if len(v) > 1 else v[0] * 3.0 / 8.0)
            sd = expert_imbalance(sd, *imb)
        model.load_state_dict(sd)
    model.eval().to(dev)
    broadcast_state_dict(model, src=0)       # weight replication over xGMI (RCCL broadcast)
    model.set_compute_dtype(dtype)
    x = synth_input(a.batch, a.imgsz, a.imgsz, seed=1 + rank).to(dev)   # resident in HBM before timing

    MAX_DET = 300
    CONF, IOU, MULTI = (0.001, 0.7, True) if a.dense else (0.25, 0.7, False)
    nms_kw = dict(max_det=MAX_DET, multi_label=MULTI, cluster=a.cluster, sigma=a.sigma)
    P = max(1, a.pipeline)

    class Slot:
        """One batch in flight: its own stream, packed result buffer (per sub-batch: dets | idx | counts), gather buffer, side streams and
        captured graph.  split: the batch walked as that many sub-batches on parallel streams inside the one graph.  nms=False: the
        forward pass alone (the synchronised leg reports forward-only next to forward + NMS)."""

        def __init__(self, split, own_stream, nms=True):
            assert a.batch % split == 0, "--split / --sync-split must divide --batch"
            self.split, self.nms, self.sub = split, nms, a.batch // split
            self.sub_words = ops.nms_pack_numel(self.sub, MAX_DET)
            self.xs = list(x.split(self.sub))
            self.stream = torch.cuda.Stream(device=dev) if own_stream else None
            self.pack = torch.empty((split * self.sub_words,), dtype=torch.float32, device=dev)
            self.gathered = torch.empty((world, self.pack.numel()), dtype=torch.float32, device=dev) if distributed else None
            self.side = [torch.cuda.Stream(device=dev) for _ in range(split - 1)]
            self.graph, self.static_out, self.gathered_ev = None, None, None

        def sub_step(self, i):
            y, _ = model._predict_once(self.xs[i])
            if not self.nms:
                return y
            return nms_padded(y, CONF, IOU, pack=self.pack[i * self.sub_words:(i + 1) * self.sub_words], **nms_kw)

        def local_step(self):
            """Forward + NMS of this rank's images.  split S: S sub-batches, the first on the current stream and the others on side
            streams forked from / joined into it (inside a captured graph these become parallel branches), each with its own slice of
            the packed result buffer."""
            cur = torch.cuda.current_stream()
            for st in self.side:
                st.wait_stream(cur)
            outs = [None] * self.split
            for i, st in enumerate(self.side, start=1):
                with torch.cuda.stream(st):
                    outs[i] = self.sub_step(i)
            outs[0] = self.sub_step(0)
            for st in self.side:
                cur.wait_stream(st)
            if not self.nms:
                return outs
            dets, counts, idx = ops.nms_pack_views(self.pack.view(self.split, self.sub_words), self.sub, MAX_DET)
            return dets, counts, idx, outs[0][3]

        def finish(self, local_out):
            """What follows the rank-local work of a step: for N>1 ONE RCCL all_gather of the packed results (dets | idx | counts,
            written in place by the NMS kernels), outside the captured graph: a collective is not part of the rank-local launch
            sequence.  Every rank issues its collectives on ONE stream in step order (the launching stream), whatever the pipeline depth."""
            if not self.nms:
                return local_out
            dets, counts, idx, status = local_out
            if distributed:
                dets, counts, idx = ops.nms_pack_views(gather_packed(self.pack, out=self.gathered).view(world, self.split, self.sub_words),
                                                       self.sub, MAX_DET)
            return dets, counts, status

        def capture(self):
            g = torch.cuda.CUDAGraph()
            # N > 1: the process group's watchdog thread polls events while this thread captures — capture errors are scoped to
            # the capturing thread there (hipStreamCaptureModeThreadLocal), so that its calls cannot invalidate the capture
            kw = {"capture_error_mode": "thread_local"} if distributed else {}
            if self.stream is not None:
                kw["stream"] = self.stream
            with torch.cuda.graph(g, **kw):
                self.static_out = self.local_step()
            self.graph = g
            g.replay()
            torch.cuda.synchronize()

        def run(self):
            """Launch one step on this slot; returns (start, end) events of its rank-local work."""
            main = torch.cuda.current_stream()
            if self.stream is None:
                e0 = main.record_event(torch.cuda.Event(enable_timing=True))
                if self.graph is not None:
                    self.graph.replay()
                local = self.static_out if self.graph is not None else self.local_step()
                e1 = main.record_event(torch.cuda.Event(enable_timing=True))
                self.finish(local)
                return e0, e1
            st = self.stream                                       # (the caller synchronises the device before its first launch: the slot
            if self.gathered_ev is not None:                       #  stream must NOT wait for the launching stream here — that stream waits for the previous step)
                st.wait_event(self.gathered_ev)                    # this slot's previous results have been consumed
            with torch.cuda.stream(st):
                e0 = st.record_event(torch.cuda.Event(enable_timing=True))
                if self.graph is not None:
                    self.graph.replay()
                    local = self.static_out
                else:
                    local = self.local_step()
                e1 = st.record_event(torch.cuda.Event(enable_timing=True))
            main.wait_event(e1)
            self.finish(local)                                     # the collective (N > 1) stays on the launching stream, in step order
            self.gathered_ev = main.record_event()
            return e0, e1

    slots = [Slot(a.split, P > 1) for _ in range(P)]
    from yolo_master_amd.options import OPTIONS
    if (OPTIONS.detect_level_streams or OPTIONS.detect_early_levels) and a.sync_split > 1:
        # Detect's side-stream switches (YMK_ENABLE 16 / 32: A/B options, off by default) fork streams inside the walk; two sub-batch walks of one
        # capture each forking their own crash the capture on ROCm 7.2 (segmentation fault inside hipStreamEndCapture) — one walk for that leg
        print("[bench] YMK_ENABLE side-stream switch: the synchronised leg uses one sub-batch (--sync-split 1)", file=sys.stderr)
        a.sync_split = 1
    sync_slots = [] if a.no_sync_leg else [Slot(a.sync_split, False), Slot(a.sync_split, False, nms=False)]

    def max_over_ranks(v: float) -> float:
        if not distributed:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    with torch.inference_mode():
        main_stream = torch.cuda.current_stream()
        for sl in slots + sync_slots:
            for _ in range(max(a.warmup // P, 1) if sl in slots else 2):
                if sl.stream is not None:
                    with torch.cuda.stream(sl.stream):
                        local = sl.local_step()
                    main_stream.wait_stream(sl.stream)
                else:
                    local = sl.local_step()
                # The collective (N > 1) is issued on the LAUNCHING stream, as in the timed loop — never on a slot's stream: that stream is
                # captured below, and the process group's watchdog thread polls the completion event of every collective it has not yet
                # retired (every ~100 ms); an event whose stream is capturing by then fails its query (hipErrorCapturedEvent) and takes
                # the process down.  First seen when this path ran on RCCL at all (round 4; profiles/r04_closures.log).
                sl.finish(local)
            torch.cuda.synchronize()
        model.check_flags()
        if distributed:
            # ... and the warm-up collectives must be RETIRED by the process group before any stream starts capturing — bounded, not timed:
            # every gather's Work handle is waited for (`gather_packed` keeps them: dist.pending_works), the device is synchronised, and a
            # barrier (itself a collective on the launching stream, waited for the same way) lines the ranks up.  A completed Work whose
            # event has been queried successfully is dropped from the watchdog's list at its next pass; the event it would query is complete
            # (not "captured") from here on, whatever the pass period is.
            from yolo_master_amd.dist import drain_collectives
            drain_collectives(dev)
        graph = None
        if not a.no_graph:   # the rank-local step (forward + NMS) is one captured HIP graph at every N (one per slot)
            try:
                for sl in slots + sync_slots:
                    sl.capture()
                graph = slots[0].graph
            except Exception as e:  # pragma: no cover
                if rank == 0:
                    print(f"[bench] HIP graph capture unavailable ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
                for sl in slots + sync_slots:
                    sl.graph, sl.static_out = None, None
                graph = None
                torch.cuda.synchronize()

        # clocks: the chip ramps for tens of milliseconds after a host-side pause (graph capture) — untimed replays until 0.3 s of
        # device work have run, so that a short timed region (the driver's 20 steps = 0.1 s) is not measured on the ramp
        t_spin = time.perf_counter()
        while graph is not None:      # (eager launches = profiling runs: the step count stays what the flags say)
            for sl in slots:
                sl.run()
            torch.cuda.synchronize()
            # (N > 1: every rank must issue the same number of gathers — the decision to stop is rank 0's... i.e. the slowest rank's)
            if max_over_ranks(1.0 if time.perf_counter() - t_spin <= 0.3 else 0.0) == 0.0:
                break

        # ---------------------------------------------------------------- the timed region: exactly K steps between barriers
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs = []
        host_us = []
        for i in range(a.steps):
            t1 = time.perf_counter()
            evs.append(slots[i % P].run())
            host_us.append((time.perf_counter() - t1) * 1e6)   # host time to ENQUEUE one step (graph replay, event records, the gather)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        p50_latency_ms = _p50([e0.elapsed_time(e1) for e0, e1 in evs])          # launch -> completion of a step's rank-local work
        # completion to completion, over windows of P steps (with several batches in flight completions come in bursts)
        per_step = [evs[i][1].elapsed_time(evs[i + P][1]) / P for i in range(a.steps - P)]
        p50_ms = _p50(per_step) if per_step else p50_latency_ms / P
        model.check_flags()                       # device flag words of the timed steps, read once after the loop

        # ---------------------------------------------------------------- synchronised leg: the reference's own convention
        # benchmarks/suite.py:316-330: synchronise, run ONE batch, synchronise, wall-clock; throughput = bs * 1000 / p50_ms; warm-up 10,
        # >= 30 measured (reports/issues-52...md:135).  One batch in flight, forward + NMS (+ gather for N > 1) and forward only.
        sync = None
        if sync_slots:
            sync = {}
            n_meas = max(30, a.steps)
            for tag, sl in (("forward_nms", sync_slots[0]), ("forward_only", sync_slots[1])):
                ts = []
                for i in range(10 + n_meas):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    sl.run()
                    torch.cuda.synchronize()
                    if i >= 10:
                        ts.append((time.perf_counter() - t1) * 1e3)
                sync[tag] = max_over_ranks(_p50(ts))
            model.check_flags()

        # roofline leg: per-call HIP events around every op family (eager launches on this stream); the object
        # reported is the family with the largest share of the step, the rest go into "families"
        roof, fams, retained, n_calls, roof_layers, roof_step = None, None, None, None, None, None
        if rank == 0:
            try:
                ops.TIMER.start()
                for _ in range(EAGER_STEPS):
                    y_, _ = model._predict_once(x)   # the whole batch on ONE stream (per-op events; no collective outside the lock-step timed loop)
                    nms_padded(y_, CONF, IOU, **nms_kw)
                torch.cuda.synchronize()
                recs = ops.TIMER.records
                shapes = list(getattr(ops.TIMER, "shapes", []))
                ops.TIMER.stop()
                n_calls = len(recs) // EAGER_STEPS
                # per-call time = MEDIAN over the eager steps (one 3-step mean was 2-3 x off on two families of config 5 in round 5)
                med_ms = [_p50([recs[i + r * n_calls][1].elapsed_time(recs[i + r * n_calls][2]) for r in range(EAGER_STEPS)]) for i in range(n_calls)]
                # routed (image, expert) pairs per ES-MoE layer of this batch (of B x top_k possible): what the expert stages' bytes scale with
                retained = {f"model.{i}": int((m.last_route["gate_w"] > 0).sum()) for i, m in enumerate(model.model)
                            if getattr(m, "last_route", None) and "gate_w" in m.last_route}
                if os.environ.get("YMK_BENCH_CALLS"):   # every op call of one step with its shape, time, GB/s and TFLOP/s (diagnostics)
                    n1 = n_calls
                    rows = []
                    for i in range(n1):
                        ms = med_ms[i]
                        rows.append((ms, i, recs[i][0], shapes[i] if i < len(shapes) else "", recs[i][3], recs[i][4]))
                    with open(os.environ["YMK_BENCH_CALLS"], "w") as fh:
                        fh.write(f"{n1} op calls, {sum(r[0] for r in rows):.3f} ms\n")
                        for ms, i, fam, shp, nb, fl in sorted(rows, reverse=True):
                            fh.write(f"{i:4d} {fam[:58]:58s} {shp:40s} {ms * 1e3:8.1f} us {nb / ms / 1e6:8.0f} GB/s {fl / ms / 1e9:8.1f} TF/s\n")
                agg = {}
                for i, (fam, e0, e1, nb, fl) in enumerate(recs[:n_calls]):   # one step's calls, each at its median time
                    r = agg.setdefault(fam, [0.0, 0, 0, 0])
                    r[0] += med_ms[i]; r[1] += nb; r[2] += fl; r[3] += 1
                ridge = MFMA_PEAK_TFLOPS[a.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)

                def describe(fam):
                    ms, nbytes, flops, n = agg[fam]
                    gbs, tfl = nbytes / (ms * 1e-3) / 1e9, flops / (ms * 1e-3) / 1e12
                    if flops / max(nbytes, 1) < ridge:
                        r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None}
                    else:
                        r = {"bound": "mfma", "achieved": round(tfl, 2), "peak": MFMA_PEAK_TFLOPS[a.dtype],
                             "unit": "TFLOP/s", "frac": round(tfl / MFMA_PEAK_TFLOPS[a.dtype], 4), "traffic": None}
                    r.update(kernel=fam, launches_per_step=n, avg_launch_us=round(ms * 1e3 / n, 2),
                             ms_per_step=round(ms, 4), alg_bytes_per_launch=int(nbytes / n),
                             alg_gflop_per_launch=round(flops / n / 1e9, 3), achieved_gbs=round(gbs, 1),
                             achieved_tflops=round(tfl, 2))
                    r["traffic"] = pmc_traffic(fam)
                    sq = sq_utilisation(fam)
                    if sq:
                        r["sq"] = sq
                    if fam in VALU_FAMILIES:
                        # stencil kernels have no matrix contraction: neither HBM nor MFMA is their limit; the honest third axis is
                        # the fp32 VALU rate (64 T MAC/s measured on MI355X for v_fma / v_pk_fma / v_dot2c, tools/micro/valu_rate.hip)
                        r["valu"] = {"achieved": round(tfl / 2, 2), "peak": VALU_PEAK_TMACS, "unit": "TMAC/s",
                                     "frac": round(tfl / 2 / VALU_PEAK_TMACS, 4)}
                    return r

                # ---- whole-step and layer-level fractions against SURVEY 8(d)'s LAYER-FUSED ideal: each model-YAML layer reads its inputs
                # once and writes its output once, every element at the compute width (ops.TIMER.io, filled by the graph walk); the per-family
                # `roofline.frac` above counts what each KERNEL must move, which includes intermediates a layer-fused ideal does not contain
                es = 2 if a.dtype in ("bf16", "f16") else 4
                layers = list(getattr(ops.TIMER, "layers", []))
                io = dict(getattr(ops.TIMER, "io", {}))
                lay_ms, lay_fl = {}, {}
                for i, ((fam, e0, e1, nb, fl), li) in enumerate(zip(recs[:n_calls], layers)):
                    lay_ms[li] = lay_ms.get(li, 0.0) + med_ms[i]
                    lay_fl[li] = lay_fl.get(li, 0) + fl
                step_bytes = sum((i_ + o_) * es for i_, o_ in io.values())
                step_flops = sum(lay_fl.values())
                groups = {}
                for i_, m_ in enumerate(model.model):
                    tname = type(m_).__name__
                    if tname in ("ES_MOE", "A2C2f", "Detect") and i_ in io:
                        g_ = groups.setdefault(tname, {"layers": [], "ms": 0.0, "bytes": 0, "flops": 0.0})
                        g_["layers"].append(i_); g_["ms"] += lay_ms.get(i_, 0.0); g_["bytes"] += sum(io[i_]) * es; g_["flops"] += lay_fl.get(i_, 0.0)
                roof_layers = {k: {"layers": v["layers"], "ms_per_step_eager": round(v["ms"], 4), "layer_fused_bytes": int(v["bytes"]),
                                   "achieved_gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                                   "hbm_frac": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                                   "achieved_tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2),
                                   "mfma_frac": round(v["flops"] / max(v["ms"], 1e-9) / 1e9 / MFMA_PEAK_TFLOPS[a.dtype], 4)}
                               for k, v in groups.items()}
                roof_step = {"layer_fused_bytes": int(step_bytes), "flops": int(step_flops), "eager_ms_sum_of_ops": round(sum(lay_ms.values()), 4)}
                if agg:
                    order = sorted(agg, key=lambda f: -agg[f][0])
                    roof = describe(a.roofline_kernel if a.roofline_kernel in agg else order[0])
                    # `roofline.frac` / `achieved` are the LAYER-GROUP figures of the group the dominant kernel belongs to (SURVEY 8(d)'s layer-fused
                    # bytes — inputs once, output once — over the group's eager time): a kernel's own algorithmic bytes may contain an
                    # intermediate the ideal does not (moe_dw: the depthwise planes), which flatters it.  The per-kernel figures stay beside them.
                    fam_layers = sorted({li for (f_, *_), li in zip(recs[:n_calls], layers) if f_ == roof["kernel"]})
                    gname = next((k for k, v in groups.items() if fam_layers and set(fam_layers) <= set(v["layers"])), None)
                    for k in ("bound", "achieved", "peak", "unit", "frac"):
                        roof["kernel_" + k] = roof[k]
                    if gname is not None:
                        g_ = groups[gname]
                        hbm_bound = g_["flops"] / max(g_["bytes"], 1) < ridge
                        gbs_, tfl_ = g_["bytes"] / max(g_["ms"], 1e-9) / 1e6, g_["flops"] / max(g_["ms"], 1e-9) / 1e9
                        roof.update(bound="hbm" if hbm_bound else "mfma", achieved=round(gbs_ if hbm_bound else tfl_, 2),
                                    peak=HBM_PEAK_GBS if hbm_bound else MFMA_PEAK_TFLOPS[a.dtype], unit="GB/s" if hbm_bound else "TFLOP/s",
                                    frac=round((gbs_ / HBM_PEAK_GBS) if hbm_bound else (tfl_ / MFMA_PEAK_TFLOPS[a.dtype]), 4),
                                    scope=f"layer group {gname} (model layers {g_['layers']}): layer-fused bytes {int(g_['bytes'])} and "
                                          f"{g_['flops'] / 1e9:.1f} GFLOP over {g_['ms']:.4f} ms eager; kernel_* = the dominant kernel alone")
                    else:
                        roof["scope"] = "dominant kernel (its layer is not part of a routed / attention / head layer group)"
                    fams = [{k: d[k] for k in ("kernel", "ms_per_step", "launches_per_step", "bound", "frac", "achieved_gbs",
                                               "achieved_tflops", "sq") if k in d} for d in map(describe, order)]
            except Exception as e:  # the throughput line must survive a failure of the diagnostic leg
                print(f"[bench] roofline leg failed ({type(e).__name__}: {e})", file=sys.stderr)
                ops.TIMER.stop()

    if rank == 0:
        total_images = world * a.batch * a.steps
        headline = a.cfg is None and (a.scale, a.batch, a.imgsz, a.dtype) == ("s", 64, 640, "bf16") and not (a.dense or a.cluster or a.imbalance)
        cfgtag = ("BASELINE.json configs[2]" + ("/[3]" if world > 1 else "")) if headline else "not the headline configuration"
        nms_tag = f"NMS conf {CONF} IoU {IOU}" + (" multi_label" if MULTI else "") + (f" + CW-NMS sigma {a.sigma}" if a.cluster else "")
        res = {
            "metric": "images/sec @ 640x640 bs=64, YOLO-Master-S; per-image p50 latency" if headline else
                      f"images/sec @ {a.imgsz}x{a.imgsz} bs={a.batch}, {a.cfg or 'YOLO-Master-' + a.scale.upper()} scale {a.scale} {a.dtype} "
                      "(diagnostic run, not BASELINE's metric)",
            "value": round(total_images / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            # `value`: K steps between the barriers with `pipeline_depth` batches in flight.  `value_sync`: ONE batch at a time,
            # synchronised wall-clock per batch, bs * 1000 / p50 — the reference's own convention (benchmarks/suite.py:316-330)
            "pipeline_depth": P,
            "value_sync": round(world * a.batch * 1e3 / sync["forward_nms"], 2) if sync else None,
            "p50_batch_ms_sync": round(sync["forward_nms"], 4) if sync else None,
            "p50_ms_per_image_sync": round(sync["forward_nms"] / a.batch, 5) if sync else None,
            "forward_only_sync": {"images_per_s": round(world * a.batch * 1e3 / sync["forward_only"], 2),
                                  "p50_batch_ms": round(sync["forward_only"], 4)} if sync else None,
            "p50_inter_completion_ms_per_image": round(p50_ms / a.batch, 5),     # inverse throughput of the pipelined region, not a latency
            "p50_batch_latency_ms": round(p50_latency_ms, 4),                    # launch -> completion of one batch INSIDE the pipelined region
            # p50 host wall time spent enqueueing one step (graph replay + events + the result gather for N > 1), rank 0: what bounds scaling once
            # the device work is sharded — 8 ranks each pay this for their own 64 images, in parallel processes
            "host_us_per_step_launch": round(_p50(host_us), 1),
            "config": {"workload": (f"YOLO-Master-{a.scale.upper()} forward+NMS, synthetic {a.imgsz}x{a.imgsz}, "
                                    f"bs={a.batch}/GPU, ES-MoE top-k=2 ({cfgtag})") if a.cfg is None else
                                   f"{a.cfg} at scale {a.scale} forward+NMS, synthetic {a.imgsz}x{a.imgsz}, bs={a.batch}/GPU (not the headline configuration)",
                       "global_batch": world * a.batch, "imgsz": a.imgsz, "parallelism": f"dp{world} (image shards, no data-path collective)",
                       "launch": ("hipGraph" if graph is not None else "eager") + (f", {a.split} sub-batches on parallel streams" if a.split > 1 else "")
                                 + (f", {P} batches in flight" if P > 1 else ""),
                       "sync_launch": None if not sync else ("hipGraph" if graph is not None else "eager") + f", one batch in flight, {a.sync_split} sub-batches on parallel streams",
                       "nms": nms_tag, "imbalance": None if not a.imbalance else a.imbalance,
                       # the reference's own convention (benchmarks/suite.py:316-330: ONE synchronised batch at a time) — kept here as well
                       # as at top level because the driver's record keeps `config` / `roofline` and drops unknown top-level keys
                       "pipeline_depth": P,
                       "value_sync": round(world * a.batch * 1e3 / sync["forward_nms"], 2) if sync else None,
                       "p50_batch_ms_sync": round(sync["forward_nms"], 4) if sync else None,
                       "forward_only_sync": {"images_per_s": round(world * a.batch * 1e3 / sync["forward_only"], 2),
                                             "p50_batch_ms": round(sync["forward_only"], 4)} if sync else None,
                       "collectives": ("RCCL (nccl) process group: weight broadcast + one packed all_gather per step" + (" — forced at world size 1" if world == 1 else ""))
                                      if distributed else None,
                       "weights": (f"cfg/cond_{a.scale}.npz over synth_state_dict(seed=0): the state_dict tests/test_gpu_baseline_configs.py::test_config3_* "
                                   "pins to the reference (input: synth_input(seed=1), the fixture's)") if use_cond else
                                  "seeded recipe + BN calibration (SURVEY 8(d); rounds 1-4's timed weights)",
                       "parity_bar": ("bf16/f16: at least as close to the reference's fp32 result as the reference's OWN run in that format (routing agreement, score / box "
                                      "percentiles x 1.25, kept-set Jaccard - 0.02: tests/golden/ref16_s640_b64.npz); fp32 (--dtype f32, config 2's type): indices / classes "
                                      "exact, scores and boxes 1e-4") if use_cond else "recipe weights: 3 x the reference's own fp32 noise + 1e-4, Jaccard >= 0.9 (tests/test_gpu_model.py)"},
            "op_calls_per_step": n_calls, "retained_pairs": retained,
            "roofline": roof,
            # whole step / layer groups against SURVEY 8(d)'s layer-fused bytes and the model's flops (not the per-kernel algorithmic bytes of
            # `roofline`): hbm_frac = bytes / time / 8 TB/s, mfma_frac = flops / time / dense peak; counter_gb = fabric traffic of one eager step
            "roofline_step": None if roof_step is None else {
                **roof_step,
                "hbm_frac": round(roof_step["layer_fused_bytes"] / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS, 4),
                "mfma_frac": round(roof_step["flops"] / (elapsed / a.steps) / 1e12 / MFMA_PEAK_TFLOPS[a.dtype], 4),
                "hbm_frac_sync": round(roof_step["layer_fused_bytes"] / (sync["forward_nms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sync else None,
                "mfma_frac_sync": round(roof_step["flops"] / (sync["forward_nms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[a.dtype], 4) if sync else None,
                "counter": step_counter_traffic()},
            "roofline_layers": roof_layers,
            "families": fams,
            "cpu_baseline": None,
        }
        if res["roofline"] is not None:
            res["roofline"]["step"] = {k: res["roofline_step"][k] for k in ("hbm_frac", "mfma_frac", "hbm_frac_sync", "mfma_frac_sync")} \
                if res["roofline_step"] else None
            res["roofline"]["value_sync"] = res["value_sync"]
            res["roofline"]["p50_batch_ms_sync"] = res["p50_batch_ms_sync"]
            # (the driver's record keeps `roofline` whole and drops unknown top-level keys: what a reader of that record needs is repeated here)
            res["roofline"]["op_calls_per_step"] = res["op_calls_per_step"]
            res["roofline"]["layers"] = None if not res["roofline_layers"] else {
                k: {q: v[q] for q in ("ms_per_step_eager", "hbm_frac", "mfma_frac")} for k, v in res["roofline_layers"].items()}
            res["roofline"]["counter_gb_per_step"] = None if not (res["roofline_step"] and res["roofline_step"]["counter"]) else round(
                res["roofline_step"]["counter"]["fetch_gb"] + res["roofline_step"]["counter"]["write_gb"], 2)
            res["roofline"]["layer_fused_gb_per_step"] = None if not res["roofline_step"] else round(res["roofline_step"]["layer_fused_bytes"] / 1e9, 3)
        if world == 1 and not a.no_cpu_baseline and a.cfg is None:
            try:
                res["cpu_baseline"] = cpu_baseline(a.scale)
            except Exception as e:  # never lose the measured line to the host-side baseline
                print(f"[bench] cpu_baseline failed ({type(e).__name__}: {e})", file=sys.stderr)
        line = json.dumps(res)
    # The JSON line is the ONLY thing the job writes to stdout.  RCCL prints a five-line version banner through C stdio when a communicator is
    # created; with stdout a pipe it sits in the C buffer until the process exits — i.e. it used to land AFTER the line (seen with --force-dist),
    # once per rank.  main() therefore points file descriptor 1 at stderr for the whole run; here every rank flushes its C stdio (to stderr), the
    # ranks meet, the process group goes away, and rank 0 prints the line on the real stdout.
    _flush_c_stdio()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(line, flush=True)


if __name__ == "__main__":
    main()
