"""GPU box: ymk_mlp_fused alone on the detector's two ABlock shapes (102 400 tokens x 128 -> 256 -> 128; 25 600 x 256 -> 512 -> 256), HIP-event
time per call; the caller sets YMK_MLP_NW8 (csrc/mlp.hip: 0 four waves with resident weights, 1 / 2 eight waves on 256 / 512 workgroups, 3 four
waves, one tile per workgroup, weights per phase).  Checks the result against the fp32 composition."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "../.."))
from yolo_master_amd import ops  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "tree"
dev = torch.device("cuda")
for (M, C, Hd) in ((102400, 128, 256), (25600, 256, 512)):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(1, 1, M, C, generator=g).to(torch.bfloat16).to(dev)
    w1 = (torch.randn(Hd, C, generator=g) * C ** -0.5).to(torch.bfloat16).to(dev)
    w2 = (torch.randn(C, Hd, generator=g) * Hd ** -0.5).to(torch.bfloat16).to(dev)
    b1, b2 = (torch.randn(Hd, generator=g) * 0.2).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    y = torch.empty_like(x)
    fn = lambda: ops.mlp_fused(x, w1, b1, w2, b2, out=y)
    fn()
    torch.cuda.synchronize()
    xs = x[0, 0, :512].float()
    ref = xs + F.silu(xs @ w1.float().t() + b1).to(torch.bfloat16).float() @ w2.float().t() + b2
    err = float((y[0, 0, :512].float() - ref).abs().max())
    for _ in range(5):
        fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{tag:24s} M {M} C {C}: {sorted(ts)[3]:7.1f} us   max err {err:.2e}", flush=True)
