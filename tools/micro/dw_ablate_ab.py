"""GPU box: where the ES-MoE depthwise stage's time goes (tools/micro/dw_ablate_ab.sh built the stage-ablated libraries): us per call at the
S detector's layers 3 and 6 with the parity-pinned weights' pair counts, median of 5 interleaved rounds."""
import ctypes as C
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
dev, bf = "cuda", torch.bfloat16
vp = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
NAMES = {0: "full", 1: "no global loads", 3: "no loads, no LDS staging writes", 4: "one filter row instead of K", 8: "no output stores", 15: "skeleton"}
libs = {a: C.CDLL(str(ROOT / "tools" / "micro" / "_dwab" / f"libdw_abl{a}.so")) for a in NAMES}
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(call, n=10):
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for Cc, HW, pairs in [(128, 160, 107), (256, 80, 98)]:
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
    out = torch.zeros(B * 2, HW, HW, Cc, device=dev, dtype=bf)
    ts = {a: [] for a in libs}
    for _ in range(5):
        for a, lib in libs.items():
            ts[a].append(timeit(lambda lib=lib: lib.ymk_esmoe_dw(1, vp(x), B, HW, HW, Cc, Cc, vp(dw_w), vp(dw_off), vp(ksz), E, top_k, 9, vp(sel), vp(dummy), vp(dummy), vp(out), s)))
    print(f"moe_dw C{Cc} @{HW}x{HW} pairs {pairs}: " + "; ".join(f"{NAMES[a]} {sorted(ts[a])[2]:.1f}" for a in libs))
