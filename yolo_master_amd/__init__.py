"""yolo_master_amd — MI355X-native (gfx950) detection forward pass of YOLO-Master.

One hot path, built from scratch behind the reference's module / model-YAML API:
conv backbone -> ES-MoE -> area-attention -> Detect decode -> batched NMS (+CW-NMS), as
hand-written HIP kernels in ``libymk.so`` (C-ABI: include/ymk.h), driven from Python.
"""
from .errors import MoERouterError, ShapeMismatchError  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):  # lazy: importing the package must not require torch/libymk
    if name in ("DetectionModel", "parse_model"):
        from .nn import tasks

        return getattr(tasks, name)
    if name in ("enable", "disable", "register_backend", "unregister_backend", "backend_class"):   # drop-in hooks under the reference's own YOLO / DetectionModel / AutoBackend objects
        from . import dropin

        return getattr(dropin, name)
    if name == "non_max_suppression":
        from .nms import non_max_suppression

        return non_max_suppression
    raise AttributeError(name)
