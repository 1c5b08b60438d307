"""GPU box, reference staged: why does one detection of the hooked segment predict differ from the un-hooked reference's?"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("YMK_REFERENCE", str(ROOT / ".refstage"))
import numpy as np
import torch
from tests.test_gpu_dropin_reference import _yolo
import yolo_master_amd
from yolo_master_amd.weights import synth_input
from yolo_master_amd.nms import non_max_suppression as ymk_nms

x = synth_input(2, 256, 256, seed=44)
kw = dict(conf=0.002, iou=0.7, verbose=False, device=0)
mref = _yolo(task="seg")
ref = mref.predict(x, **kw)
m = _yolo(task="seg")
yolo_master_amd.enable(m)
got = m.predict(x, **kw)
yolo_master_amd.disable(m)
out = {}
for i, (g, r) in enumerate(zip(got, ref)):
    out[f"got{i}"] = g.boxes.data.float().cpu().numpy(); out[f"ref{i}"] = r.boxes.data.float().cpu().numpy()
# same head output through both NMS implementations
from ultralytics.utils import nms as rnms
net = mref.model.to("cuda:0").eval()
with torch.inference_mode():
    p = net(x.to("cuda:0"))
y = p
while isinstance(y, (list, tuple)):
    y = y[0]
print("head output", tuple(y.shape))
nc = len(net.names)
a = rnms.non_max_suppression(y.clone(), 0.002, 0.7, nc=nc, max_det=300)
b = ymk_nms(y.clone().float().contiguous(), 0.002, 0.7, nc=nc, max_det=300)
for i in range(len(a)):
    ai, bi = a[i].float().cpu(), b[i].float().cpu()
    out[f"nms_ref{i}"] = ai.numpy(); out[f"nms_ymk{i}"] = bi.numpy()
    n = min(len(ai), len(bi))
    d = (ai[:n, :6] - bi[:n, :6]).abs().max(1).values
    bad = torch.nonzero(d > 1e-3).reshape(-1)
    print(f"image {i}: ref kept {len(ai)} ymk kept {len(bi)}; rows differing position-wise: {bad[:10].tolist()} (of {len(bad)})")
    for j in bad[:3].tolist():
        print("  ref", ai[j, :6].tolist()); print("  ymk", bi[j, :6].tolist())
out["y"] = y.float().cpu().numpy()
np.savez_compressed(ROOT / "gpurun_out" / "seg_dbg.npz", **out)
for i in range(2):
    g, r = out[f"got{i}"], out[f"ref{i}"]
    print(f"predict image {i}: got {g.shape} ref {r.shape}")
    n = min(len(g), len(r))
    d = np.abs(g[:n, :4] - r[:n, :4]).max(1)
    idx = np.nonzero(d > 0.02)[0]
    print("  position-wise box diffs > 0.02:", idx[:12].tolist(), "of", len(idx))
    for j in idx[:4]:
        print("   ref", r[j].tolist()); print("   got", g[j].tolist())
