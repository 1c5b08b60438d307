"""ORACLE helper: make the real reference importable in the build container.

`ultralytics` (the reference) imports cv2 / torchvision metadata at import time
(ultralytics/utils/__init__.py:24,79); neither is installed here.  This shim installs a
permissive stub `cv2` and patches importlib.metadata.version for torchvision, then puts
/root/reference on sys.path.  It is used ONLY by tests/golden/make_golden.py and by tests that
are skipped when /root/reference is absent (the GPU box never has it).
"""
import importlib.machinery
import importlib.metadata as md
import os
import sys
import types
from unittest import mock

REF = os.environ.get("YMK_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "ultralytics"))


def boot():
    if not available():
        raise RuntimeError(f"reference checkout not found at {REF}")
    if "ultralytics" in sys.modules:
        return

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return mock.MagicMock(name=f"{self.__name__}.{k}")

    cv2 = _Any("cv2")
    cv2.__spec__ = importlib.machinery.ModuleSpec("cv2", None)
    cv2.__version__ = "4.10.0"
    cv2.IMREAD_COLOR = 1
    sys.modules["cv2"] = cv2
    _v = md.version
    md.version = lambda n: "0.25.0" if n == "torchvision" else _v(n)
    os.environ.setdefault("YOLO_CONFIG_DIR", "/tmp")
    sys.path.insert(0, REF)


def stub_torchvision():
    """`predict()` / `val()` of the reference import torchvision (nn/autobackend.py:324-325, engine/predictor.py:274) and, once it
    is in sys.modules, use `torchvision.ops.nms` (utils/nms.py:156-161).  Not installed here: a stub whose `ops.nms` is the
    reference's own pure-torch TorchNMS.nms, i.e. the path the reference takes without torchvision."""
    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    tv.__spec__ = importlib.machinery.ModuleSpec("torchvision", None)
    tv.__version__ = "0.25.0"
    tvops = types.ModuleType("torchvision.ops")
    tv.ops = tvops
    sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, tvops
    from ultralytics.utils.nms import TorchNMS

    tvops.nms = TorchNMS.nms
