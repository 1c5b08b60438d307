"""GPU A/B: depthwise convolutions of the S detector at batch 64 (bf16) on the matrix-core kernel (csrc/dwmfma.hip) vs the
VALU stencil (csrc/dwconv.hip), per shape, HIP-event timed inside one process.    python tools/gpu_dw_ab.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from yolo_master_amd import ops  # noqa: E402

DEV = "cuda:0"
bf = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    print("plain depthwise (bias + residual), 64 images:   shape, VALU us, MFMA us, speed-up, MFMA GB/s (read x + write y)")
    for H, C, k in ((40, 128, 7), (20, 256, 7), (80, 128, 3), (40, 256, 3), (40, 128, 3), (20, 256, 3), (20, 128, 3), (160, 128, 9)):
        x = torch.randn(64, H, H, C, generator=g).to(bf).to(DEV)
        w = (torch.randn(C, 1, k, k, generator=g) / k).to(DEV)
        wp = ops.pack_dw_weight(w, bf)
        wp.toeplitz = ops.dw_toeplitz(wp, k, force=True)
        plain = wp.clone()
        b = torch.randn(C, generator=g).to(DEV)
        out = torch.empty_like(x)
        tv = timeit(lambda: ops.dwconv2d(x, plain, b, k, True, out=out, residual=x))
        tm = timeit(lambda: ops.dwconv2d(x, wp, b, k, True, out=out, residual=x))
        print(f"  C{C:4d} k{k} @{H:3d}^2   {tv:8.1f} {tm:8.1f}   x{tv / tm:5.2f}   {3 * x.numel() * 2 / tm / 1e3:7.0f} GB/s")
    print("ES-MoE depthwise stage (experts 3/5/7/9, two experts per image, 64 images):   shape, VALU us, MFMA us, speed-up")
    for H, C in ((160, 128), (80, 256), (40, 256), (20, 512)):
        B, E, top_k = 64, 4, 2
        x = torch.randn(B, H, H, C, generator=g).to(bf).to(DEV)
        ks = [3, 5, 7, 9]
        wps = [ops.pack_dw_weight((torch.randn(C, 1, k, k, generator=g) / k).to(DEV), bf) for k in ks]
        dw_w = torch.cat([w.reshape(-1) for w in wps])
        offs, o = [], 0
        for w in wps:
            offs.append(o)
            o += w.numel()
        toep = torch.cat([ops.dw_toeplitz(w, k, force=True) for w, k in zip(wps, ks)])
        sel = torch.tensor([[b % 4, (b + 1 + b // 4) % 4] for b in range(B)], dtype=torch.int32)
        sel, _ = sel.sort(1)
        pairs = [[] for _ in range(E)]
        for b in range(B):
            for s in range(top_k):
                pairs[int(sel[b, s])].append(b * top_k + s)
        off = [0]
        for e in range(E):
            off.append(off[-1] + len(pairs[e]))
        i32 = dict(dtype=torch.int32, device=DEV)
        ksd, offd = torch.tensor(ks, **i32), torch.tensor(off, **i32)
        flat = torch.tensor([p for e in range(E) for p in pairs[e]], **i32)
        seld, dwoff = sel.to(DEV), torch.tensor(offs, **i32)
        tv = timeit(lambda: ops.esmoe_dw(x, dw_w, dwoff, ksd, 9, top_k, seld, offd, flat))
        tm = timeit(lambda: ops.esmoe_dw(x, dw_w, dwoff, ksd, 9, top_k, seld, offd, flat, toep=toep, kmask=30))
        a = ops.esmoe_dw(x, dw_w, dwoff, ksd, 9, top_k, seld, offd, flat).float()
        m = ops.esmoe_dw(x, dw_w, dwoff, ksd, 9, top_k, seld, offd, flat, toep=toep, kmask=30).float()
        print(f"  C{C:4d} @{H:3d}^2   {tv:8.1f} {tm:8.1f}   x{tv / tm:5.2f}   max |valu - mfma| {float((a - m).abs().max()):.3e} (max |y| {float(a.abs().max()):.2f})")


if __name__ == "__main__":
    with torch.inference_mode():
        main()
