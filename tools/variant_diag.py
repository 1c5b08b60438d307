"""Which specialised kernel family moves the fp32 result?  Runs the test_gpu_variants script under several
YMK_DISABLE masks (one interpreter each) and prints max |dy| against the all-generic run."""
import os, subprocess, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests.test_gpu_variants import SCRIPT
tmp = Path(tempfile.mkdtemp())
def run(mask):
    out = tmp / f"y_{mask}.npz"
    r = subprocess.run([sys.executable, "-c", SCRIPT, str(ROOT), str(out), "f32"], env=dict(os.environ, YMK_DISABLE=str(mask)),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    return np.load(out)["y"]
ref = run(0xFFFF)
for name, bit in (("conv_stream(ws+tile)", 1), ("moe_stream", 2), ("nms_sort", 4), ("stem_mfma", 8), ("res_prefetch", 16), ("stem_rows", 32)):
    y = run(0xFFFF ^ bit)
    d = np.abs(y - ref)
    print(f"enable only {name:22s}: max|dy| boxes {d[:, :4].max():.3e} scores {d[:, 4:].max():.3e}  worst image {int(d.reshape(64, -1).max(1).argmax())}")
