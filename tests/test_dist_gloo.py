"""N>1 path on CPU: world_size-2 gloo processes exercise sharding, weight broadcast and result gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolo_master_amd.dist import shard_range


def test_shard_range_partition():
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from yolo_master_amd.dist import broadcast_state_dict, gather_detections, init_from_env
    from yolo_master_amd.nn.modules import C3k2

    r, _, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)            # different weights per rank before the broadcast
    m = C3k2(16, 32, 1, False, 0.25)
    for p in m.parameters():
        torch.nn.init.normal_(p)
    broadcast_state_dict(m, src=0)
    flat = torch.cat([t.reshape(-1).float() for t in m.state_dict().values()])
    ref = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(ref, flat)
    same = all(torch.equal(ref[0], t) for t in ref)
    # result gather: rank r owns images [b0, b1) of a batch of 6, padded detections
    b0, b1 = shard_range(6, rank, world)
    dets = torch.zeros((b1 - b0, 4, 6))
    counts = torch.zeros((b1 - b0,), dtype=torch.int32)
    for i, b in enumerate(range(b0, b1)):
        counts[i] = b % 4
        dets[i, : b % 4, 4] = float(b)
    gd, gc, _ = gather_detections(dets, counts)
    ok = gc.tolist() == [b % 4 for b in range(6)] and all(float(gd[b, 0, 4]) == (b if b % 4 else 0.0) for b in range(6))
    q.put((rank, same, ok))
    dist.destroy_process_group()


def _run_two_process(port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=120) for _ in procs]
    except Exception:
        res = None
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    return res, [p.exitcode for p in procs]


def test_two_process_gloo():
    # the rendezvous port is picked by the OS and released before the workers bind it: retry on the rare collision
    last = None
    for _ in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        res, codes = _run_two_process(port)
        last = (res, codes)
        if res is not None and all(c == 0 for c in codes):
            assert all(same and ok for _, same, ok in res), res
            return
    raise AssertionError(f"two-process gloo run failed three times: {last}")


def _worker8(rank, world, port, n_images, q):
    """One rank of a world-size-8 step at BASELINE config 4's shard sizes: images [b0, b1) of the batch, results written into ONE
    packed buffer (dets | idx | counts, ops.nms_pack_views) of the COMMON shard size (the last shards of an uneven batch are padded
    with empty images), one all_gather of it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from yolo_master_amd import ops
    from yolo_master_amd.dist import gather_packed, init_from_env

    init_from_env("gloo")
    max_det = 300
    b_common = -(-n_images // world)                      # ceil: every rank gathers the same number of words
    b0, b1 = shard_range(n_images, rank, world)
    pack = torch.zeros((ops.nms_pack_numel(b_common, max_det),), dtype=torch.float32)
    dets, counts, idx = ops.nms_pack_views(pack, b_common, max_det)
    for i, b in enumerate(range(b0, b1)):                 # image b "detects" (b % 7) boxes carrying its own index
        n = b % 7
        counts[i] = n
        dets[i, :n, 4] = float(b)
        dets[i, :n, 5] = float(b % 80)
        idx[i, :n] = torch.arange(n, dtype=torch.int32) + b
    out = None
    for _ in range(2):                                    # the pre-allocated output is reused by the second step
        g = gather_packed(pack, out=out)
        assert out is None or g.data_ptr() == out.data_ptr()
        out = g
    gd, gc, gi = ops.nms_pack_views(g, b_common, max_det)
    ok = tuple(gd.shape) == (world, b_common, max_det, 6)
    for r in range(world):
        r0, r1 = shard_range(n_images, r, world)
        for i in range(b_common):
            b = r0 + i
            want = b % 7 if b < r1 else 0                 # padding images of a short shard hold nothing
            ok &= int(gc[r, i]) == want
            if want:
                ok &= float(gd[r, i, 0, 4]) == float(b) and float(gd[r, i, want - 1, 5]) == float(b % 80) and int(gi[r, i, want - 1]) == b + want - 1
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _run_world(world, n_images):
    last = None
    for _ in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker8, args=(r, world, port, n_images, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=300) for _ in procs]
        except Exception:
            res = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        last = (res, [p.exitcode for p in procs])
        if res is not None and all(c == 0 for c in last[1]):
            return res
    raise AssertionError(f"world-size-{world} gloo run failed three times: {last}")


def test_eight_ranks_bs512_one_packed_gather():
    """BASELINE config 4's partition: 512 images over 8 ranks = 64 per rank (shard_range(512, r, 8)), one packed all_gather per step."""
    assert [shard_range(512, r, 8) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]
    res = _run_world(8, 512)
    assert len(res) == 8 and all(ok for _, ok in res), res


def test_eight_ranks_uneven_batch_padded_last_shards():
    """500 images over 8 ranks: shards of 63 / 62 images, every rank gathers the common 63 (the short shards pad with empty images)."""
    sizes = [e - b for b, e in (shard_range(500, r, 8) for r in range(8))]
    assert sizes == [63] * 4 + [62] * 4
    res = _run_world(8, 500)
    assert len(res) == 8 and all(ok for _, ok in res), res
