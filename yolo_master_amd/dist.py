"""Multi-GPU batched inference: one process per GPU, images sharded, no data-path collective.

The detection forward pass has no exchange step (ES-MoE routing is per image, BatchNorm is in eval mode,
NMS is per image), so a batch is split into contiguous per-rank shards — the same contiguous
split as the reference's DDP validation sampler (ultralytics/data/build.py:150 ContiguousDistributedSampler).
Only two RCCL collectives exist, both outside the forward pass: a one-time weight ``broadcast`` from rank 0
(replication over xGMI) and a per-batch fixed-size ``all_gather`` of the padded detections
(``[B_local, max_det, 6]`` + counts: <= 7.2 KB per image — latency-bound, never the per-link-bound ring
all-reduce).  Backend "nccl" is RCCL on ROCm; the same code runs on gloo/CPU tensors for the tests.
"""
from __future__ import annotations

import os

import collections

import torch
import torch.distributed as dist

_PENDING = collections.deque(maxlen=256)   # Work handles of the most recent result gathers (nccl), for drain_collectives


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [begin, end) shard of n_items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def forced() -> bool:
    """YMK_DIST_FORCE=1: run the multi-GPU code path (process group, weight broadcast, packed result gather) at world size 1 too —
    the smoke test of the RCCL branch on a box with one GPU (bench.py --force-dist, tests/test_gpu_bench_flow.py)."""
    return os.environ.get("YMK_DIST_FORCE", "") not in ("", "0")


def _active() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("YMK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_state_dict(module: torch.nn.Module, src: int = 0) -> None:
    """Replicate rank `src`'s parameters and buffers to every rank (one flat broadcast per dtype)."""
    if not _active():
        return
    tensors = [t for t in module.state_dict().values() if torch.is_tensor(t)]
    by_dtype: dict = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, ts in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        if dist.get_backend() == "gloo" and flat.is_cuda:
            host = flat.cpu()
            dist.broadcast(host, src=src)
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    if hasattr(module, "repack"):
        module.repack()


def gather_packed(pack: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """ONE collective per step: all_gather of every rank's packed NMS result (ops.nms_pack_numel words: dets | idx | counts, written
    in place by the NMS kernels — nothing is copied to form it).  pack: float32 [n] with the same n on every rank (pad the last shard
    of an uneven batch to the common B_local).  Returns float32 [world, n] in rank order = the original image order for contiguous
    shards (ops.nms_pack_views(result, B_local, max_det) gives dets [world, B_local, max_det, 6], counts, idx).
    out: the caller's pre-allocated [world, n] buffer (a serving loop reuses it every batch)."""
    if not _active():
        return pack.unsqueeze(0)
    world = dist.get_world_size()
    if out is None or tuple(out.shape) != (world, pack.numel()) or out.dtype != pack.dtype or out.device != pack.device:
        out = torch.empty((world, pack.numel()), dtype=pack.dtype, device=pack.device)
    if dist.get_backend() == "nccl":      # RCCL over xGMI: 0.54 MB per rank at 64 images x 300 detections — latency-bound
        w = dist.all_gather_into_tensor(out, pack, async_op=True)
        w.wait()                          # stream-level: the current stream waits for the collective (what async_op=False does), the host does not
        while _PENDING and _PENDING[0].is_completed():   # a serving loop that never drains does not accumulate retired handles
            _PENDING.popleft()
        _PENDING.append(w)                # kept (bounded) so that drain_collectives can wait for completion instead of sleeping
        return out
    src = pack.cpu()                      # gloo (CPU tests / single-GPU functional runs)
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    out.copy_(torch.stack(parts, 0))
    return out


def drain_collectives(device=None, timeout_s: float = 30.0) -> int:
    """Bounded replacement for "sleep and hope" before a HIP-graph capture on a rank that has issued collectives: every Work handle this
    module still holds is waited for, the device is synchronised, completion is CONFIRMED (`is_completed()` = the Work's end event queried
    successfully; polled with a deadline, not a fixed pause), then a barrier lines the ranks up and is itself synchronised.  After this no
    collective of this process is in flight, so the process group's watchdog has only completed events left to query — none of them on a
    stream that is about to capture (collectives are only ever issued on the launching stream, bench.py).  Returns the number of Works
    drained.  Raises TimeoutError when a Work does not complete within `timeout_s` (a hung peer), instead of capturing into a live queue."""
    import time

    if not _active():
        return 0
    works = list(_PENDING)
    _PENDING.clear()
    for w in works:
        w.wait()
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)
    deadline = time.monotonic() + timeout_s
    for w in works:
        while not w.is_completed():
            if time.monotonic() > deadline:
                raise TimeoutError("a warm-up collective did not complete before graph capture")
            time.sleep(0.001)
    dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)
    return len(works)


def gather_detections(dets: torch.Tensor, counts: torch.Tensor, idx: torch.Tensor | None = None, out: dict | None = None):
    """all_gather of per-rank padded detections.  dets [B_local, max_det, 6], counts [B_local].
    Every rank must hold the same B_local (pad the last shard).  Returns tensors with leading dim
    world*B_local in rank order (i.e. the original batch order for contiguous shards).
    out: a dict the caller keeps across batches — the gathered tensors are allocated in it on first use and reused
    afterwards (a serving loop gathers into the same three buffers every batch: no allocator traffic per step)."""
    if not _active():
        return dets, counts, idx
    world = dist.get_world_size()

    def ag(name, t):
        shape = (world * t.shape[0], *t.shape[1:])
        buf = None if out is None else out.get(name)
        if buf is None or tuple(buf.shape) != shape or buf.dtype != t.dtype or buf.device != t.device:
            buf = torch.empty(shape, dtype=t.dtype, device=t.device)
            if out is not None:
                out[name] = buf
        if dist.get_backend() == "nccl":  # RCCL: one fused all_gather straight into the output tensor
            dist.all_gather_into_tensor(buf, t.contiguous())
            return buf
        src = t.contiguous().cpu()        # gloo (CPU tests / single-GPU functional runs)
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        buf.copy_(torch.cat(parts, 0))
        return buf

    return ag("dets", dets), ag("counts", counts), (ag("idx", idx) if idx is not None else None)
