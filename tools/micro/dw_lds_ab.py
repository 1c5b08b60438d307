"""GPU box: A/B of the depthwise stencil's LDS forms (tools/micro/dw_lds_ab.sh built them): ES-MoE depthwise stage at the S detector's four
layers with the parity-pinned weights' pair counts, and the plain 7 x 7 / 3 x 3 stencils of the ABlocks / Detect; outputs compared bitwise
(the three forms run the same arithmetic in the same order), timings interleaved over ROUNDS rounds, median reported."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
dev, bf = "cuda", torch.bfloat16
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timeit(call, n=10):
    for _ in range(2):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


libs = {m: C.CDLL(str(ROOT / "tools" / "micro" / "_dwab" / f"libdw_mode{m}.so")) for m in (0, 1, 2, 3)}
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
print(f"# us per call, median of {ROUNDS} interleaved rounds; mode 0 = rounds 1-4 (ds_read2_b64), 1 = un-paired ds_read_b64, 2 = planar ds_read_b128, 3 = planar pixel pairs + v_dot2")
for Cc, HW, pairs in [(128, 160, 107), (256, 80, 98), (256, 40, 105), (512, 20, 92)]:
    B, E, top_k, ks = 64, 4, 2, [3, 5, 7, 9]
    x = torch.randn(B, HW, HW, Cc, device=dev).to(bf)
    parts, offs, off = [], [], 0
    for k in ks:
        w = (torch.randn(k * k, Cc, device=dev) / k).to(bf)
        parts.append(w.reshape(-1)); offs.append(off); off += w.numel()
    dw_w = torch.cat(parts)
    dw_off = torch.tensor(offs, dtype=torch.int32, device=dev)
    ksz = torch.tensor(ks, dtype=torch.int32, device=dev)
    sel = torch.full((B, 2), -1, dtype=torch.int32)
    g = torch.Generator().manual_seed(1)
    for b in range(B):
        p = torch.randperm(4, generator=g)
        sel[b, 0] = p[0]
        if b < pairs - B:
            sel[b, 1] = p[1]
    sel = sel.to(dev)
    dummy = torch.zeros(8, dtype=torch.int32, device=dev)
    outs = {m: torch.zeros(B * 2, HW, HW, Cc, device=dev, dtype=bf) for m in libs}
    calls = {m: (lambda m=m: libs[m].ymk_esmoe_dw(1, vp(x), B, HW, HW, Cc, Cc, vp(dw_w), vp(dw_off), vp(ksz), E, top_k, 9, vp(sel), vp(dummy), vp(dummy), vp(outs[m]), s)) for m in libs}
    ts = {m: [] for m in libs}
    for _ in range(ROUNDS):
        for m in libs:
            ts[m].append(timeit(calls[m]))
    same = all(torch.equal(outs[0], outs[m]) for m in (1, 2))
    live = (sel >= 0).reshape(-1)
    d3 = (outs[3].float() - outs[0].float())[live]
    print(f"moe_dw C{Cc} @{HW}x{HW} pairs {int((sel >= 0).sum())}: " + "  ".join(f"mode{m} {med(ts[m]):7.1f}" for m in libs) + f"  modes 0-2 bit-identical {same}; "
          f"mode 3 vs 0: {float((d3 != 0).float().mean()) * 100:.3f} % of the outputs differ, max |d| {float(d3.abs().max()):.2e}")
for Cc, HW, k, res in [(128, 40, 7, True), (256, 20, 7, True), (128, 80, 3, False), (256, 40, 3, False)]:
    B = 64
    x = torch.randn(B, HW, HW, Cc, device=dev).to(bf)
    r = torch.randn(B, HW, HW, Cc, device=dev).to(bf) if res else None
    w = (torch.randn(k * k, Cc, device=dev) / k).to(bf)
    bias = torch.randn(Cc, device=dev)
    outs = {m: torch.zeros(B, HW, HW, Cc, device=dev, dtype=bf) for m in libs}
    calls = {m: (lambda m=m: libs[m].ymk_dwconv2d(1, vp(x), vp(w), vp(bias), vp(r), vp(outs[m]), B, HW, HW, Cc, k, Cc, Cc, Cc if res else 0, 0 if res else 1, s)) for m in libs}
    ts = {m: [] for m in libs}
    for _ in range(ROUNDS):
        for m in libs:
            ts[m].append(timeit(calls[m], 20))
    same = all(torch.equal(outs[0], outs[m]) for m in (1, 2))
    d3 = outs[3].float() - outs[0].float()
    print(f"dwconv C{Cc} k{k} @{HW}x{HW}{' +res' if res else ' +silu'}: " + "  ".join(f"mode{m} {med(ts[m]):7.1f}" for m in libs) + f"  modes 0-2 bit-identical {same}; "
          f"mode 3 vs 0: {float((d3 != 0).float().mean()) * 100:.3f} % differ, max |d| {float(d3.abs().max()):.2e}")
