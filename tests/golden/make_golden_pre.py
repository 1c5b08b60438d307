"""Golden vectors for the pre-processing oracle: the REAL reference's `LetterBox.get_params` (ultralytics/data/augment.py:1752-1800)
over a sweep of image shapes and option sets -> tests/golden/pre_params.json.  get_params is pure arithmetic (it never calls
cv2), so it runs under the stub cv2 of oracle/refboot.py.  Run in the build container:  python tests/golden/make_golden_pre.py"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle import pre_ref, refboot  # noqa: E402

refboot.boot()
from ultralytics.data.augment import LetterBox  # noqa: E402

SHAPES = [(1080, 810), (720, 1280), (640, 640), (480, 640), (375, 500), (1, 1), (17, 4000), (3000, 11), (641, 639), (1279, 1281),
          (320, 320), (1280, 1280), (333, 777), (427, 640), (500, 333), (96, 160), (2160, 3840), (101, 203), (1281, 1279), (639, 641)]
OPTS = [dict(), dict(auto=True), dict(scaleup=False), dict(center=False), dict(scale_fill=True), dict(new_shape=(384, 672)),
        dict(new_shape=(1280, 1280), auto=True, stride=64), dict(new_shape=(320, 320), scaleup=False, center=False)]

rows, bad = [], 0
for opt in OPTS:
    for shape in SHAPES:
        lb = LetterBox(**{"new_shape": (640, 640), **opt})
        p = lb.get_params({"img": np.zeros((*shape, 3), np.uint8)})
        rec = {"shape": list(shape), "opt": {k: (list(v) if isinstance(v, tuple) else v) for k, v in opt.items()},
               "new_unpad": [int(v) for v in p["new_unpad"]], "top": int(p["top"]), "bottom": int(p["bottom"]), "left": int(p["left"]),
               "right": int(p["right"]), "ratio": [float(v) for v in p["ratio"]]}
        o = pre_ref.letterbox_params(shape, **{"new_shape": (640, 640), **opt})
        same = (list(o["new_unpad"]) == rec["new_unpad"] and (o["top"], o["bottom"], o["left"], o["right"]) ==
                (rec["top"], rec["bottom"], rec["left"], rec["right"]) and list(o["ratio"]) == rec["ratio"])
        bad += not same
        rows.append(rec)
json.dump(rows, open(HERE / "pre_params.json", "w"))
print(f"{len(rows)} parameter sets from the real reference; oracle restatement identical on {len(rows) - bad}")
assert bad == 0
