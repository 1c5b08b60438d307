"""GPU box: the qkv projection inside the attention kernel (ymk_area_attn_qkv) against the two launches it replaces (128 -> 384 1x1 convolution +
ymk_area_attn) on the detector's 40^2 A2C2f shape (64 images, 4 heads, area 4), HIP-event time per call; argv[1] = library (default: the tree's)."""
import os
import sys

import torch

if len(sys.argv) > 1:
    import shutil
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "../../yolo_master_amd/libymk.so")
    shutil.copy(here, "/tmp/libymk_keep.so")
    shutil.copy(sys.argv[1], here)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "../.."))
from yolo_master_amd import ops  # noqa: E402

tag = sys.argv[2] if len(sys.argv) > 2 else "tree"
dev = torch.device("cuda")
B, H, heads, area = (64, 40, 4, 4) if os.environ.get("QKV_SHAPE", "40") == "40" else (64, 20, 8, 1)
C = heads * 32
g = torch.Generator().manual_seed(1)
x = torch.randn(B, H, H, C, generator=g).to(torch.bfloat16).to(dev)
w = ops.pack_conv_weight((torch.randn(3 * C, C, 1, 1, generator=g) * C ** -0.5), torch.bfloat16).to(dev)
b = (torch.randn(3 * C, generator=g) * 0.2).to(dev)
o, v = ops.new_act(B, H, H, C, x.dtype, dev), ops.new_act(B, H, H, C, x.dtype, dev)
qkv = ops.new_act(B, H, H, 3 * C, x.dtype, dev)


def unfused():
    ops.conv2d(x, w, b, 1, 1, False, out=qkv)
    ops.area_attn(qkv, heads, area, out=o)


def fused():
    ops.area_attn_qkv(x, w, b, heads, area, out=o, v_out=v)


def timeit(fn):
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
    return sorted(ts)[3]


unfused()
ref_o, ref_v = o.clone(), qkv[..., 2 * C:].clone()
fused()
torch.cuda.synchronize()
print(f"{tag:20s} [{H}^2 C{C}] conv + area_attn {timeit(unfused):7.1f} us   area_attn_qkv {timeit(fused):7.1f} us   "
      f"max |o - o'| {float((o.float() - ref_o.float()).abs().max()):.2e}  v differs in {float((v != ref_v).float().mean()):.4f}", flush=True)
if len(sys.argv) > 1:
    shutil.copy("/tmp/libymk_keep.so", here)
