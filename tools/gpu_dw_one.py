"""One depthwise shape, a few launches, on one path — the workload of a rocprofv3 --pmc pass (tools/gpu_dw_pmc.sh).
    python tools/gpu_dw_one.py <H> <C> <k> <mfma|valu> [iters]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from yolo_master_amd import ops  # noqa: E402

H, C, k, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
with torch.inference_mode():
    x = torch.randn(64, H, H, C, generator=g).to(bf).to("cuda:0")
    wp = ops.pack_dw_weight((torch.randn(C, 1, k, k, generator=g) / k).to("cuda:0"), bf)
    if mode == "valu":
        wp = wp.clone()
    out = torch.empty_like(x)
    for _ in range(iters):
        ops.dwconv2d(x, wp, None, k, False, out=out)
    torch.cuda.synchronize()
