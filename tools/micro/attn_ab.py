"""GPU box: ymk_area_attn alone on the detector's two A2C2f shapes (64 images; 40^2 x 128 channels, 4 heads, area 4 and 20^2 x 256, 8 heads,
area 1), HIP-event time per launch, for ONE library (argv[1], default the tree's) — the caller loops over libraries / environment values
(tools/micro/attn_ab.sh).  Also checks the result against a torch fp32 softmax(q k^T / sqrt(d)) v of the same 16-bit operands."""
import ctypes as C
import os
import sys

import torch

lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "../../yolo_master_amd/libymk.so"))
tag = sys.argv[2] if len(sys.argv) > 2 else "tree"
dev = torch.device("cuda")
for (B, H, heads, area) in ((64, 40, 4, 4), (64, 20, 8, 1)):
    Cq, N = heads * 32, H * H
    g = torch.Generator().manual_seed(H)
    qkv = (torch.randn(B, N, 3 * Cq, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    out = torch.zeros(B, N, Cq, dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.ymk_area_attn(1, C.c_void_p(qkv.data_ptr()), 3 * Cq, C.c_void_p(out.data_ptr()), Cq, B, N, heads, area, C.c_void_p(st))
    assert call() == 0
    torch.cuda.synchronize()
    Na = N // area
    q, k, v = [t.float().view(4, area, Na, heads, 32).permute(0, 1, 3, 2, 4) for t in qkv[:4].split(Cq, dim=2)]  # noqa
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, -1) @ v).permute(0, 1, 3, 2, 4).reshape(4, N, Cq)
    err = float((out[:4].float() - ref).abs().max())
    for _ in range(5):
        call()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ts.sort()
    print(f"{tag:28s} {H}^2 heads {heads} area {area}: {ts[3]:7.1f} us  (min {ts[0]:.1f})  max err vs fp32 {err:.2e}", flush=True)
