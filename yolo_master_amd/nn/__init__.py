"""Module library + model builder (reference surface: ultralytics.nn)."""
from .modules import *  # noqa: F401,F403
from .tasks import DetectionModel, parse_model, yaml_model_load  # noqa: F401
