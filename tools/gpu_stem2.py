"""GPU box: time ymk_stem_pair at the benchmark shape (64 x 3 x 640 x 640, bf16) — median of event-timed launches.
YMK_DISABLE=4096 times the variant with the stem on the fp32 matrix cores.  (Stage ablation of the first version, 64 images:
all 431 us; without the stem's fp32 MFMAs 325, without its SiLU 390, without its LDS gathers 402, without the next-tile prefetch
398, without row 1's MFMAs 431: the fp32 matrix-core stem was a quarter of the kernel, hence the bf16-split default.)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from yolo_master_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
x = torch.rand(B, 3, 640, 640, generator=g).cuda()
w0 = (torch.randn(32, 27, generator=g) * 0.3).cuda()
b0 = (torch.randn(32, generator=g) * 0.3).cuda()
w1 = ops.pack_conv_weight(torch.randn(64, 32, 3, 3, generator=g) * 288 ** -0.5, torch.bfloat16).cuda()
b1 = (torch.randn(64, generator=g) * 0.2).cuda()
wt0 = w0.t().contiguous()
y = ops.stem_pair(x, wt0, b0, w1, b1)
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.stem_pair(x, wt0, b0, w1, b1, out=y); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"stem_pair B={B}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us")
